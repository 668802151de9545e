/*
 * ra_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-message-at-a-time restatement of the rabbitmq/ra per-server Raft
 * transition (reference app vsn 3.1.10) for the message classes on the ra_gpu_batch hot
 * path.  It is the CPU checker the HIP path is compared with: only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it.  The product library
 * (libra_gpu_batch.so) never links, loads or calls anything in this directory.
 *
 * The reference is Erlang and no Erlang/OTP toolchain exists in this environment, so the
 * reference itself cannot be executed here (no oracle/_ref).  This restatement is pinned by
 * the known-answer vectors transcribed from the reference's own unit tests
 * (tests/golden/ra_server_suite_vectors.json <- test/ra_server_SUITE.erl,
 * src/ra_server.erl:4225-4238, test/ra_log_2_SUITE.erl); clauses no reference test pins are
 * listed "source-derived" in DESIGN.md.
 *
 * Deliberately a different formulation from the device kernel: the log is an explicit
 * per-index array of terms (one fetch_term per entry, as ra_log does), the quorum is a
 * real sort, every clause is written in reference order.  All citations are
 * /root/reference/<path>:<lines>.
 */
#include "ra_oracle.h"

#include <stdlib.h>
#include <string.h>

#define UNDEF RGB_UNDEF

/* ------------------------------------------------------------------ log model ---- */
/* ra_log state: range (undefined or {first,last}), last_term, last_written_index_term,
 * current snapshot; terms[] holds the term of every index in the range
 * (src/ra_log.erl:96-130 record fields range/last_term/last_written_index_term). */
typedef struct {
  int      has_range;
  uint64_t first, last;
  uint64_t last_term;
  uint64_t lw_idx, lw_term;
  uint64_t snap_idx, snap_term;   /* UNDEF = no snapshot */
  uint64_t pend_first;            /* pending (src/ra_log.erl:126): indexes handed to the WAL and not yet
                                     confirmed, always the contiguous tail [pend_first .. last] on this
                                     path; empty is held canonically as last_index + 1 */
  uint64_t base;                  /* index stored at terms[0] */
  uint64_t *terms;
  size_t   cap;
  /* sparse pending (after ra_log:write_sparse/3, or while a two-range written event is handled): the ra_seq as
   * an EXPLICIT ascending list of indexes, every ra_seq operation done index by index the way the reference's
   * iterators do (src/ra_seq.erl:44-110, 144-160, 278-311) -- deliberately unlike the engine's interval
   * arithmetic.  NULL = the contiguous-tail form above.  Lists are never edited in place (a failed assertion
   * restores the message's starting state): pend_orig is the list the current message started with. */
  uint64_t *pend_idx;
  size_t   pend_n;
  uint64_t *pend_orig;
} olog;

typedef struct {
  uint64_t current_term, commit_index, last_applied;
  uint64_t cond_reply[4];
  uint64_t match_index[RGB_MAX_MEMBERS], next_index[RGB_MAX_MEMBERS],
           commit_index_sent[RGB_MAX_MEMBERS];
  uint8_t  role, cond_reason, self, n_members, voted_for, leader_id, votes;
  uint8_t  present_mask, voter_mask, status_mask, self_nonvoter, cond_leader;
  uint8_t  backoff_mask;           /* peers whose status is {snapshot_backoff, _} (never normal at the same time) */
  uint64_t pre_vote_token;
  uint32_t machine_version, effective_machine_version;
  uint64_t query_index;                       /* src/ra_server.erl:96 */
  uint64_t peer_query_index[RGB_MAX_MEMBERS]; /* ra_peer_state().query_index src/ra.hrl:61-73 */
} oscal;

typedef struct {
  oscal s;
  olog  log;
} oserver;

struct ora_ctx {
  uint32_t n_servers, n_members;
  uint32_t max_pipeline_count, max_aer_batch;
  uint32_t max_runs;              /* 0 = unbounded; else the engine's term-run table size (ora_set_max_runs) */
  oserver *sv;
};

static int log_reserve(olog *l, uint64_t idx) {
  /* make terms[idx - base] addressable */
  if (l->terms == NULL) {
    l->cap = 64;
    l->terms = (uint64_t *)malloc(l->cap * sizeof(uint64_t));
    if (!l->terms) return -1;
    l->base = idx;
  }
  if (idx < l->base) {
    size_t shift = (size_t)(l->base - idx);
    size_t ncap = l->cap + shift;
    uint64_t *nt = (uint64_t *)malloc(ncap * sizeof(uint64_t));
    if (!nt) return -1;
    memcpy(nt + shift, l->terms, l->cap * sizeof(uint64_t));
    free(l->terms);
    l->terms = nt; l->cap = ncap; l->base = idx;
  }
  if (idx - l->base >= l->cap) {
    size_t ncap = l->cap;
    while (idx - l->base >= ncap) ncap *= 2;
    uint64_t *nt = (uint64_t *)realloc(l->terms, ncap * sizeof(uint64_t));
    if (!nt) return -1;
    l->terms = nt; l->cap = ncap;
  }
  return 0;
}

/* ra_log:fetch_term/2, src/ra_log.erl:1186-1200: defined only for Idx in range
 * (?IS_IN_RANGE, src/ra_log.erl:477-480), else undefined. */
static uint64_t log_fetch_term(const olog *l, uint64_t idx) {
  if (l->has_range && idx >= l->first && idx <= l->last)
    return l->terms[idx - l->base];
  return UNDEF;
}

/* ra_log:last_index_term/1, src/ra_log.erl:830-835 */
static void log_last_index_term(const olog *l, uint64_t *idx, uint64_t *term) {
  if (l->has_range) { *idx = l->last; *term = l->last_term; }
  else { *idx = l->snap_idx; *term = l->snap_term; }
}

/* pending = ra_seq of [pend_first .. range last]; canonical empty form */
static int log_pend_nonempty(const olog *l) { return l->has_range && l->pend_first <= l->last; }
static void log_pend_canon(olog *l) {
  if (!log_pend_nonempty(l)) {
    uint64_t li, lt;
    log_last_index_term(l, &li, &lt);
    l->pend_first = li + 1;
  }
}

/* ---- sparse pending: literal ra_seq operations on an explicit index list ---- */
static void pend_replace(olog *l, uint64_t *nv, size_t n) {
  if (l->pend_idx && l->pend_idx != l->pend_orig) free(l->pend_idx);
  l->pend_idx = nv; l->pend_n = n;
}
/* pend_first <- the start of the list's trailing run when that run ends at the last index; the list is dropped
 * once it is nothing but that run (or empty): back to the contiguous-tail form */
static void pend_sync(olog *l) {
  uint64_t li, lt;
  log_last_index_term(l, &li, &lt);
  size_t n = l->pend_n, k = n;
  if (n && l->has_range && l->pend_idx[n - 1] == li) {
    k = n - 1;
    while (k > 0 && l->pend_idx[k - 1] + 1 == l->pend_idx[k]) k--;
    l->pend_first = l->pend_idx[k];
  } else {
    l->pend_first = li + 1;
  }
  if (k == 0) pend_replace(l, NULL, 0);                     /* no index below the trailing run */
}
static int pend_materialize(olog *l) {                      /* contiguous-tail form -> explicit list */
  if (l->pend_idx) return 0;
  size_t n = log_pend_nonempty(l) ? (size_t)(l->last - l->pend_first + 1) : 0;
  uint64_t *v = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
  if (!v) return -1;
  for (size_t i = 0; i < n; i++) v[i] = l->pend_first + i;
  pend_replace(l, v, n);
  return 0;
}
/* a fresh list: the elements of the current one that satisfy lo <= x <= hi, followed by extra[0..n_extra) */
static int pend_filter_append(olog *l, uint64_t lo, uint64_t hi, uint64_t app_first, uint64_t app_n) {
  uint64_t *v = (uint64_t *)malloc((l->pend_n + app_n + 1) * sizeof(uint64_t));
  if (!v) return -1;
  size_t n = 0;
  for (size_t i = 0; i < l->pend_n; i++)
    if (l->pend_idx[i] >= lo && l->pend_idx[i] <= hi) v[n++] = l->pend_idx[i];
  for (uint64_t i = 0; i < app_n; i++) v[n++] = app_first + i;   /* ra_seq:append/2 per index (Idx > last) */
  pend_replace(l, v, n);
  return 0;
}

/* ra_log:next_index/1, src/ra_log.erl:1166-1174 */
static uint64_t log_next_index(const olog *l) {
  if (l->has_range) return l->last + 1;
  if (l->snap_idx != UNDEF) return l->snap_idx + 1;
  return 0;
}

/* ra_log:exists/2, src/ra_log.erl:1459-1467 */
static int log_exists(const olog *l, uint64_t idx, uint64_t term) {
  uint64_t t = log_fetch_term(l, idx);
  return t != UNDEF && t == term;
}

/* ra_server:fetch_term/2 with the snapshot fallback, src/ra_server.erl:3185-3196 */
static uint64_t srv_fetch_term(const olog *l, uint64_t idx) {
  uint64_t t = log_fetch_term(l, idx);
  if (t == UNDEF) {
    if (l->snap_idx != UNDEF && l->snap_idx == idx) return l->snap_term;
    return UNDEF;
  }
  return t;
}

enum { HLE_OK = 0, HLE_MISMATCH = 1, HLE_MISSING = 2 };
/* has_log_entry_or_snapshot/3, src/ra_server.erl:3168-3183 */
static int has_log_entry_or_snapshot(const olog *l, uint64_t idx, uint64_t term) {
  uint64_t t = log_fetch_term(l, idx);
  if (t == UNDEF) {
    if (l->snap_idx != UNDEF && l->snap_idx == idx)
      return l->snap_term == term ? HLE_OK : HLE_MISMATCH;
    return HLE_MISSING;
  }
  return t == term ? HLE_OK : HLE_MISMATCH;
}

/* entry term of the k-th entry (0-based) of an AER message */
static uint64_t msg_entry_term(const rgb_msg *m, uint32_t k) {
  return k < m->n_run0 ? m->run0_term : m->run1_term;
}
static uint64_t msg_first_index(const rgb_msg *m) { return m->a + 1 + (uint64_t)m->gap; }

/* ra_log:write/2, src/ra_log.erl:547-599 + wal_write_batch range update :1618-1623.
 * Writes entries k0..n-1 of message m.  Returns 0 or an RGB_INV_* code; validates first,
 * mutates only on success (Erlang terms are immutable: a crash leaves the old state). */
static int log_write(olog *l, const rgb_msg *m, uint32_t k0) {
  uint64_t fst = msg_first_index(m) + k0;
  uint64_t lst = msg_first_index(m) + (m->n_entries - 1);
  if (l->has_range && !(fst <= l->last + 1))              /* guard :557-559 */
    return RGB_INV_WRITE_INTEGRITY;
  if (fst == 0) return RGB_INV_WRITE_INTEGRITY;            /* index 0 is never rewritten */
  /* NewRange = ra_range:new(Start, LastIdx) :1617-1622, and new/2 with Start > End is `undefined`
   * (src/ra_range.erl:41-50): a write that ends below the start of a sparse range leaves NO range.  What follows
   * reads last_index_term/1 :831-835 -- the snapshot's -- so a log without a snapshot cannot go there */
  const int range_lost = l->has_range && l->first > lst;
  if (range_lost && l->snap_idx == UNDEF) return RGB_INV_WRITE_INTEGRITY;
  /* overwrite lowers last_written, :565-581 */
  uint64_t lwi = fst - 1 < l->lw_idx ? fst - 1 : l->lw_idx;
  uint64_t lwt;
  if (lwi == l->lw_idx) {
    lwt = l->lw_term;
  } else if (l->snap_idx != UNDEF && l->snap_idx == lwi) {
    lwt = l->snap_term;
  } else if (lwi == 0) {                                   /* _ when LWIdx =< 0 */
    lwt = 0;
  } else {
    lwt = log_fetch_term(l, lwi);
    if (lwt == UNDEF) return RGB_INV_LAST_WRITTEN_TERM;    /* true = Term2 =/= undefined */
  }
  if (log_reserve(l, fst) || log_reserve(l, lst)) return RGB_INV_WRITE_INTEGRITY;
  for (uint32_t k = k0; k < m->n_entries; k++)
    l->terms[msg_first_index(m) + k - l->base] = msg_entry_term(m, k);
  if (!l->has_range) { l->has_range = 1; l->first = fst; }
  l->last = lst;
  l->last_term = msg_entry_term(m, m->n_entries - 1);
  if (range_lost) l->has_range = 0;
  l->lw_idx = lwi; l->lw_term = lwt;
  /* Pend = ra_seq:limit(FstIdx - 1, Pend0) :583, then ra_seq:append per entry :1610 */
  if (l->pend_idx) {
    if (pend_filter_append(l, 0, fst - 1, fst, lst - fst + 1)) return RGB_INV_WRITE_INTEGRITY;
    pend_sync(l);
    return 0;
  }
  if (fst < l->pend_first) l->pend_first = fst;
  return 0;
}

/* ra_log:append/2 of one leader entry, src/ra_log.erl:482-545 (cursor effect only) */
static int log_append(olog *l, uint64_t idx, uint64_t term) {
  if (log_reserve(l, idx)) return -1;
  l->terms[idx - l->base] = term;
  if (!l->has_range) { l->has_range = 1; l->first = idx; }
  l->last = idx; l->last_term = term;
  if (l->pend_idx) {                                       /* limit(Idx-1) + append(Idx) :503-505 */
    if (pend_filter_append(l, 0, idx ? idx - 1 : 0, idx, 1)) return -1;
    if (idx == 0) { uint64_t *v = (uint64_t *)malloc(sizeof(uint64_t)); if (!v) return -1; v[0] = 0; pend_replace(l, v, 1); }
    pend_sync(l);
    return 0;
  }
  if (idx < l->pend_first) l->pend_first = idx;           /* limit(Idx-1) + append(Idx) :503-505 */
  return 0;
}

/* ra_log:set_last_index/2, src/ra_log.erl:842-893 */
static int log_set_last_index(olog *l, uint64_t idx) {
  uint64_t t = log_fetch_term(l, idx);
  int snap_is_idx = (l->snap_idx != UNDEF && l->snap_idx == idx);
  if (t == UNDEF && !snap_is_idx) return RGB_INV_SET_LAST_INDEX_NOT_FOUND;
  if (snap_is_idx) {
    /* range = ra_range:limit(Idx+1, Range), src/ra_range.erl:80-91 */
    if (l->has_range) {
      if (idx + 1 <= l->first) l->has_range = 0;
      else if (idx + 1 <= l->last) l->last = idx;
    }
    l->last_term = l->snap_term;
    l->lw_idx = l->snap_idx; l->lw_term = l->snap_term;
    if (l->pend_idx) { if (pend_filter_append(l, 0, idx, 0, 0)) return RGB_INV_SET_LAST_INDEX_NOT_FOUND; pend_sync(l); return 0; }
    if (idx + 1 < l->pend_first) l->pend_first = idx + 1;  /* pending = ra_seq:limit(Idx, Pend0) :868 */
    log_pend_canon(l);
    return 0;
  }
  uint64_t lwi = idx < l->lw_idx ? idx : l->lw_idx;
  uint64_t lwt;
  if (l->snap_idx != UNDEF && l->snap_idx == lwi) lwt = l->snap_term;
  else lwt = log_fetch_term(l, lwi);
  if (lwt == UNDEF) return RGB_INV_LAST_WRITTEN_TERM;      /* true = LWTerm =/= undefined */
  if (l->has_range) {
    if (idx + 1 <= l->first) l->has_range = 0;
    else if (idx + 1 <= l->last) l->last = idx;
  }
  l->last_term = t;
  l->lw_idx = lwi; l->lw_term = lwt;
  if (l->pend_idx) { if (pend_filter_append(l, 0, idx, 0, 0)) return RGB_INV_SET_LAST_INDEX_NOT_FOUND; pend_sync(l); return 0; }
  if (idx + 1 < l->pend_first) l->pend_first = idx + 1;    /* pending = ra_seq:limit(Idx, Pend0) :891 */
  log_pend_canon(l);
  return 0;
}

/* ra_log:handle_event({written, Term, Seq}), src/ra_log.erl:897-944, for a contiguous
 * Seq = [from..to].  ra_seq:remove_prefix/2 (src/ra_seq.erl:144-147, drop_prefix :278-291) on a
 * contiguous pending tail: prefix elements below the tail are skipped; a prefix that starts
 * above the tail's first index is {error, not_prefix}.  Returns 1 if last_written changed;
 * *resend is set when the reference calls resend_pending/2 (:917-919, State0 otherwise
 * unchanged: the re-send itself is WAL I/O and stays on the host); *inv when the
 * {ok, Pend} = ... match of the snapshot clause (:929) would fail. */
/* The same event for a sequence of two ranges and/or a sparse `pending`: the written sequence and `pending` as
 * explicit index lists, the retry of :931-943 one index at a time from the top of the sequence, and
 * ra_seq:remove_prefix/2 as the two-iterator walk of drop_prefix/2 (src/ra_seq.erl:278-291). */
/* (round 5) any number of ranges: `extra` = n_extra (first, last) pairs below the two inline ranges, ascending --
 * the RGB_MF_SEQX range list of the batch (rgb_submit_seq) */
static int log_written_sparse(olog *l, uint64_t term, const uint64_t *extra, uint32_t n_extra, uint64_t lo_s, uint64_t lo_e,
                              int two, uint64_t from, uint64_t to, int *resend, int *inv) {
  size_t nw = (size_t)(to - from + 1) + (two ? (size_t)(lo_e - lo_s + 1) : 0);
  for (uint32_t k = 0; k < n_extra; k++) nw += (size_t)(extra[2 * k + 1] - extra[2 * k] + 1);
  uint64_t *w = (uint64_t *)malloc((nw ? nw : 1) * sizeof(uint64_t));
  if (!w) return 0;
  size_t n = 0;
  for (uint32_t k = 0; k < n_extra; k++) for (uint64_t i = extra[2 * k]; i <= extra[2 * k + 1]; i++) w[n++] = i;
  if (two) for (uint64_t i = lo_s; i <= lo_e; i++) w[n++] = i;
  for (uint64_t i = from; i <= to; i++) w[n++] = i;
  int changed = 0;
  for (size_t top = n; top > 0; top--) {                    /* WrittenSeq = w[0..top) */
    const uint64_t idx = w[top - 1];
    const uint64_t t = log_fetch_term(l, idx);
    const int clause1 = (t != UNDEF && t == term);
    const int clause2 = (t == UNDEF && l->snap_idx != UNDEF && idx <= l->snap_idx);
    if (!clause1 && !clause2) continue;                     /* term mismatch: limit(Idx-1, Seq), retry */
    if (pend_materialize(l)) break;
    /* drop_prefix/2: P over w[0..top), S over pending */
    size_t pi = 0, si = 0, keep_from = l->pend_n;
    int not_prefix = 0;
    for (;;) {
      if (si >= l->pend_n) { keep_from = l->pend_n; break; }            /* drop_prefix(_, end_of_seq) -> {ok, []} */
      if (pi >= top) { keep_from = si; break; }                         /* prefix exhausted: the rest stays */
      if (w[pi] == l->pend_idx[si]) { pi++; si++; }
      else if (w[pi] < l->pend_idx[si]) pi++;                           /* prefix index below the sequence: skipped */
      else { not_prefix = 1; break; }
    }
    if (not_prefix) {
      pend_sync(l);                                         /* unchanged content: back to its canonical form */
      if (clause1) *resend = 1; else *inv = RGB_INV_WRITTEN_NOT_PREFIX;
      free(w);
      return 0;
    }
    uint64_t *v = (uint64_t *)malloc((l->pend_n - keep_from + 1) * sizeof(uint64_t));
    if (!v) break;
    memcpy(v, l->pend_idx + keep_from, (l->pend_n - keep_from) * sizeof(uint64_t));
    pend_replace(l, v, l->pend_n - keep_from);
    if (clause1) {
      changed = !(l->lw_idx == idx && l->lw_term == term);
      l->lw_idx = idx; l->lw_term = term;
    }
    pend_sync(l);
    free(w);
    return changed;
  }
  free(w);
  return 0;
}

static int log_written(olog *l, uint64_t term, uint64_t from, uint64_t to, int *resend, int *inv) {
  if (l->pend_idx) return log_written_sparse(l, term, NULL, 0, 0, 0, 0, from, to, resend, inv);
  uint64_t idx = to;
  for (;;) {
    uint64_t t = log_fetch_term(l, idx);
    if (t != UNDEF && t == term) {
      if (log_pend_nonempty(l)) {
        if (from > l->pend_first) { *resend = 1; return 0; }
        if (idx + 1 > l->pend_first) l->pend_first = idx + 1;
        log_pend_canon(l);
      }
      int changed = !(l->lw_idx == idx && l->lw_term == term);
      l->lw_idx = idx; l->lw_term = term;
      return changed;
    }
    if (t == UNDEF && l->snap_idx != UNDEF && idx <= l->snap_idx) {
      /* snapshot overtook the write: only pending is trimmed */
      if (log_pend_nonempty(l)) {
        if (from > l->pend_first) { *inv = RGB_INV_WRITTEN_NOT_PREFIX; return 0; }
        if (idx + 1 > l->pend_first) l->pend_first = idx + 1;
        log_pend_canon(l);
      }
      return 0;
    }
    /* term mismatch (or undefined above the snapshot): ra_seq:limit(Idx-1, Seq) and retry */
    if (idx == 0 || idx - 1 < from) return 0;
    idx -= 1;
  }
}

/* ra_log:handle_event({snapshot_written, {SnapIdx, SnapTerm}, _, snapshot, _, _}),
 * src/ra_log.erl:1054-1150: only when the range is defined and SnapIdx >= its first index;
 * last_written follows the snapshot when it is not above it; the range is truncated
 * (ra_range:truncate/2, src/ra_range.erl:93-106).  Returns 1 if last_written changed. */
static int log_snapshot_written(olog *l, uint64_t snap_idx, uint64_t snap_term) {
  if (!(l->has_range && snap_idx >= l->first)) return 0;   /* stale: second clause, no change */
  int changed = 0;
  if (!(l->lw_idx > snap_idx)) {
    changed = !(l->lw_idx == snap_idx && l->lw_term == snap_term);
    l->lw_idx = snap_idx; l->lw_term = snap_term;
  }
  if (snap_idx >= l->last) l->has_range = 0;               /* truncate: nothing left */
  else l->first = snap_idx + 1;
  l->snap_idx = snap_idx; l->snap_term = snap_term;
  /* Pend = ra_seq:floor(SmallestLiveIdx, Pend0) :1100-1107 with no live indexes below the
   * snapshot (live-index tracking is machine state and stays on the host) */
  if (l->pend_idx) { if (!pend_filter_append(l, snap_idx + 1, UNDEF, 0, 0)) pend_sync(l); return changed; }
  if (snap_idx + 1 > l->pend_first) l->pend_first = snap_idx + 1;
  log_pend_canon(l);
  return changed;
}

/* --------------------------------------------------------------- server helpers -- */
typedef struct {
  uint32_t flags;
  uint32_t invariant;
  int      has_reply;
  uint64_t r_term, r_next, r_last, r_lterm;
  uint8_t  reply_to;
  uint64_t w_first, w_last;
  rgb_rpc *rpcs; uint32_t rpc_cap, n_rpcs_total; uint8_t n_rpcs;
  uint32_t msg_index;
  int      vote_reqs;              /* {send_vote_requests,..}: request fields ride in r_* */
  uint8_t  hb_mask;                /* heartbeat_rpc_effects/4: peers that get a #heartbeat_rpc{} */
  uint8_t  cancel_mask;            /* {cancel_snapshot_retry_timer, Peer}: make_all_rpcs/1 :2356-2363 */
  uint64_t hb_term, hb_query_index;
  uint64_t q_consensus;            /* RGB_F_QUERY_QUORUM */
} ofx;

static int is_present(const oscal *s, unsigned i) {
  return i < RGB_MAX_MEMBERS && ((s->present_mask >> i) & 1u);
}

/* become(follower,_,_) resets every peer status to normal, src/ra_server.erl:2183-2192 */
static void set_role(oscal *s, uint8_t role, ofx *fx) {
  if (s->role != role) fx->flags |= RGB_F_ROLE_CHANGED;
  if (role == RGB_ROLE_FOLLOWER && s->role != RGB_ROLE_FOLLOWER) {
    s->status_mask = 0xFF;
    s->backoff_mask = 0;
  }
  if (role != RGB_ROLE_AWAIT_CONDITION) s->cond_reason = RGB_COND_NONE;
  s->role = role;
}

/* update_term_and_voted_for/3, src/ra_server.erl:3041-3058 */
static void update_term_and_voted_for(oscal *s, uint64_t term, uint8_t voted_for, ofx *fx) {
  if (term == s->current_term && voted_for == s->voted_for) return;
  fx->flags |= RGB_F_PERSIST;
  s->current_term = term;
  s->voted_for = voted_for;
  /* reset_query_index/1 :3769-3773: every cluster entry's query_index := 0 */
  memset(s->peer_query_index, 0, sizeof s->peer_query_index);
}
/* update_term/2, src/ra_server.erl:3060-3064 */
static void update_term(oscal *s, uint64_t term, ofx *fx) {
  if (term != UNDEF && term > s->current_term)
    update_term_and_voted_for(s, term, RGB_NONE, fx);
}

static void set_leader_id(oscal *s, uint8_t l, ofx *fx) {
  if (s->leader_id != l) fx->flags |= RGB_F_LEADER_CHANGED;
  s->leader_id = l;
}

/* append_entries_reply/3, src/ra_server.erl:3624-3631 */
static void aer_reply(const oserver *sv, uint64_t term, int success, uint8_t to, ofx *fx) {
  uint64_t li, lt;
  log_last_index_term(&sv->log, &li, &lt);
  fx->has_reply = 1;
  fx->flags |= RGB_F_REPLY | (success ? RGB_F_REPLY_SUCCESS : 0);
  fx->r_term = term; fx->r_next = li + 1;
  fx->r_last = sv->log.lw_idx; fx->r_lterm = sv->log.lw_term;
  fx->reply_to = to;
}

static void vote_reply(uint64_t term, int granted, uint8_t to, ofx *fx) {
  fx->has_reply = 1;
  fx->flags |= RGB_F_REPLY | RGB_F_REPLY_VOTE | (granted ? RGB_F_REPLY_SUCCESS : 0);
  fx->r_term = term; fx->r_next = 0; fx->r_last = 0; fx->r_lterm = 0;
  fx->reply_to = to;
}

/* apply_to/5, src/ra_server.erl:3250-3282: last_applied := min(LastIdx, ApplyTo) when that
 * advances it (ra_machine:apply itself runs on the host). Returns 1 when it advanced. */
static int apply_to(oserver *sv, uint64_t apply_to_idx) {
  if (apply_to_idx > sv->s.last_applied) {
    uint64_t li, lt;
    log_last_index_term(&sv->log, &li, &lt);
    uint64_t to = li < apply_to_idx ? li : apply_to_idx;
    if (to >= sv->s.last_applied + 1) { sv->s.last_applied = to; return 1; }
  }
  return 0;
}

/* evaluate_commit_index_follower/2, src/ra_server.erl:2246-2280 */
static void evaluate_commit_index_follower(oserver *sv, ofx *fx) {
  if (sv->s.leader_id == RGB_NONE) return;
  uint64_t li, lt;
  log_last_index_term(&sv->log, &li, &lt);
  uint64_t at = li < sv->s.commit_index ? li : sv->s.commit_index;
  if (apply_to(sv, at)) fx->flags |= RGB_F_APPLIED | RGB_F_AUX_EVAL;
}

static int cmp_desc(const void *a, const void *b) {
  uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return x < y ? 1 : (x > y ? -1 : 0);
}

/* agreed_commit/1, src/ra_server.erl:3684-3688 */
uint64_t ora_agreed_commit(const uint64_t *idxs, uint32_t n) {
  uint64_t tmp[64];
  if (n == 0 || n > 64) return UNDEF;
  memcpy(tmp, idxs, n * sizeof(uint64_t));
  qsort(tmp, n, sizeof(uint64_t), cmp_desc);
  uint32_t nth = n / 2 + 1;                                /* trunc(length/2) + 1, 1-based */
  return tmp[nth - 1];
}

/* match_indexes/1 :3671-3682, increment_commit_index/1 :3648-3657, evaluate_quorum/2 :3633-3646 */
static void evaluate_quorum(oserver *sv, ofx *fx) {
  oscal *s = &sv->s;
  uint64_t list[RGB_MAX_MEMBERS + 1];
  uint32_t n = 0;
  list[n++] = sv->log.lw_idx;                              /* the leader's last WRITTEN index */
  for (unsigned i = 0; i < s->n_members; i++) {
    if (i == s->self || !is_present(s, i)) continue;
    if (!((s->voter_mask >> i) & 1u)) continue;            /* non-voters excluded */
    list[n++] = s->match_index[i];
  }
  uint64_t ci0 = s->commit_index;
  uint64_t p = ora_agreed_commit(list, n);
  uint64_t t = srv_fetch_term(&sv->log, p);
  if (t != UNDEF && t == s->current_term)                  /* Raft 5.4.2; NO max(): may decrease */
    s->commit_index = p;
  if (s->commit_index > ci0) fx->flags |= RGB_F_AUX_EVAL;
  if (apply_to(sv, s->commit_index)) fx->flags |= RGB_F_APPLIED;
}

/* heartbeat_reply/2 :3727-3729 cast to the sender of the #heartbeat_rpc{} */
static void heartbeat_reply(uint64_t term, uint64_t query_index, uint8_t to, ofx *fx) {
  fx->has_reply = 1;
  fx->flags |= RGB_F_REPLY | RGB_F_REPLY_HEARTBEAT;
  fx->r_term = term; fx->r_next = query_index; fx->r_last = 0; fx->r_lterm = 0;
  fx->reply_to = to;
}

static unsigned n_peers(const oscal *s) {                  /* maps:size(peers(State)) */
  unsigned n = 0;
  for (unsigned i = 0; i < s->n_members; i++)
    if (i != s->self && is_present(s, i)) n++;
  return n;
}

/* heartbeat_rpc_effects/4 + heartbeat_rpc_effect_for_peer/5 :3775-3795: peers with status normal
 * whose query_index is below QueryIndex */
static void heartbeat_rpc_effects(const oscal *s, uint64_t query_index, ofx *fx) {
  uint8_t mask = 0;
  for (unsigned i = 0; i < s->n_members; i++) {
    if (i == s->self || !is_present(s, i)) continue;
    if (!((s->status_mask >> i) & 1u)) continue;
    if (s->peer_query_index[i] < query_index) mask |= (uint8_t)(1u << i);
  }
  if (mask) {
    fx->flags |= RGB_F_SEND_HEARTBEATS;
    fx->hb_mask |= mask;
    fx->hb_term = s->current_term; fx->hb_query_index = query_index;
  }
}

/* update_heartbeat_rpc_effects/1 :3731-3747 (the waiting queue lives on the host: with no
 * peers every waiting query applies now) */
static void update_heartbeat_rpc_effects(const oscal *s, ofx *fx) {
  if (n_peers(s) == 0) { fx->flags |= RGB_F_QUERY_APPLY; return; }
  heartbeat_rpc_effects(s, s->query_index, fx);
}

/* make_heartbeat_rpc_effects/2 :3749-3767 */
static void make_heartbeat_rpc_effects(oscal *s, ofx *fx) {
  if (n_peers(s) == 0) { fx->flags |= RGB_F_QUERY_APPLY; return; }
  s->query_index += 1;
  heartbeat_rpc_effects(s, s->query_index, fx);
  /* queue:in({NewQueryIndex, QueryRef}, Waiting): the host queues it under this index */
  fx->hb_term = s->current_term; fx->hb_query_index = s->query_index;
}

/* heartbeat_rpc_quorum/3 :3797-3814 -> update_peer_query_index/3 :3816-3829,
 * get_current_query_quorum/1 :3831-3832 over query_indexes/1 :3659-3669 */
static void heartbeat_rpc_quorum(oscal *s, uint64_t new_query_index, unsigned peer, ofx *fx) {
  if (is_present(s, peer) && peer < RGB_MAX_MEMBERS && new_query_index > s->peer_query_index[peer])
    s->peer_query_index[peer] = new_query_index;
  uint64_t list[RGB_MAX_MEMBERS + 1];
  uint32_t n = 0;
  list[n++] = s->query_index;
  for (unsigned i = 0; i < s->n_members; i++) {
    if (i == s->self || !is_present(s, i)) continue;
    if (!((s->voter_mask >> i) & 1u)) continue;            /* membership =/= voter excluded */
    list[n++] = s->peer_query_index[i];
  }
  fx->flags |= RGB_F_QUERY_QUORUM;
  fx->q_consensus = ora_agreed_commit(list, n);
}

static void emit_rpc(ofx *fx, const rgb_rpc *r) {
  if (fx->rpcs && fx->n_rpcs_total < fx->rpc_cap) fx->rpcs[fx->n_rpcs_total] = *r;
  fx->n_rpcs_total++;
  fx->n_rpcs++;
}

/* make_pipelined_rpc_effects/3 :2285-2346, make_rpc_effect/5 :2382-2416,
 * make_append_entries_rpc/6 :2418-2435.  Returns an RGB_INV_* code or 0; sets *more. */
static int make_pipelined_rpc_effects(struct ora_ctx *c, oserver *sv, uint32_t srv_id,
                                      int force, int *more, ofx *fx) {
  oscal *s = &sv->s;
  uint64_t next_log_idx = log_next_index(&sv->log);
  uint64_t max_pipe = c->max_pipeline_count, max_batch = c->max_aer_batch;
  *more = 0;
  for (unsigned i = 0; i < s->n_members; i++) {
    if (i == s->self || !is_present(s, i)) continue;
    if (!((s->status_mask >> i) & 1u)) continue;           /* status := normal */
    uint64_t ni = s->next_index[i], mi = s->match_index[i];
    if (!(ni < next_log_idx || s->commit_index_sent[i] < s->commit_index)) continue;
    /* NumInFlight = NextIdx - MatchIdx - 1 is an Erlang integer and may be negative */
    int64_t inflight = (int64_t)(ni - mi) - 1;
    if (!(inflight < (int64_t)max_pipe || force)) continue;
    int64_t room = (int64_t)max_pipe - inflight;
    int64_t bs = (int64_t)max_batch < room ? (int64_t)max_batch : room;
    if (bs < 1) bs = 1;
    /* make_rpc_effect */
    uint64_t prev = ni - 1;
    uint64_t prev_term = log_fetch_term(&sv->log, prev);
    rgb_rpc r;
    memset(&r, 0, sizeof r);
    r.msg_index = fx->msg_index; r.server = srv_id; r.peer = (uint8_t)i;
    r.term = s->current_term; r.leader_commit = s->commit_index;
    uint64_t new_ni;
    if (prev_term == UNDEF && !(sv->log.snap_idx != UNDEF && sv->log.snap_idx == prev)) {
      /* {send_snapshot,..}: next index is NOT advanced past SnapIdx, :2403-2415 */
      /* next_index 0: PrevIdx is -1 in the reference's integers, below any snapshot index */
      if (sv->log.snap_idx == UNDEF || !(ni == 0 || prev < sv->log.snap_idx))
        return RGB_INV_PIPELINE_PREV_UNDEFINED;            /* case_clause / ?assert(PrevIdx < SnapIdx) */
      r.kind = RGB_RPC_SNAPSHOT;
      r.prev_log_index = sv->log.snap_idx;
      r.prev_log_term = sv->log.snap_term;
      new_ni = sv->log.snap_idx;                           /* {SnapIdx, Effect, State} */
      fx->flags |= RGB_F_SEND_SNAPSHOT;
    } else {
      if (prev_term == UNDEF) prev_term = sv->log.snap_term;
      uint64_t li, lt;
      log_last_index_term(&sv->log, &li, &lt);
      uint64_t to = prev + (uint64_t)bs < li ? prev + (uint64_t)bs : li;
      r.kind = RGB_RPC_AER;
      r.prev_log_index = prev; r.prev_log_term = prev_term;
      r.n_entries = (uint16_t)(to >= prev + 1 ? to - prev : 0);
      new_ni = to + 1;
    }
    if (!(new_ni >= ni)) return RGB_INV_NEXT_INDEX_REGRESSED; /* ?assert(NewNextIdx >= NextIdx) */
    r.next_index = new_ni;
    emit_rpc(fx, &r);
    s->next_index[i] = new_ni;
    s->commit_index_sent[i] = s->commit_index;
    int64_t new_inflight = (int64_t)(new_ni - mi) - 1;
    if (new_ni < next_log_idx && new_inflight < (int64_t)max_pipe) *more = 1;
  }
  return 0;
}

/* required_quorum/1 :3996-3999, count_voters/1 :4001-4009 */
static unsigned required_quorum(const oscal *s) {
  unsigned voters = 0;
  for (unsigned i = 0; i < s->n_members; i++)
    if (is_present(s, i) && ((s->voter_mask >> i) & 1u)) voters++;
  return voters / 2 + 1;
}

/* the leader branch of handle_candidate(#request_vote_result{vote_granted = true}) :1055-1058 */
static void become_leader(oserver *sv, ofx *fx) {
  oscal *s = &sv->s;
  /* initialise_peers/1 :3234-3242: every member next_index = ra_log:next_index,
   * match_index 0, commit_index_sent 0, status normal */
  uint64_t ni = log_next_index(&sv->log);
  for (unsigned i = 0; i < s->n_members; i++) {
    if (!is_present(s, i)) continue;
    s->next_index[i] = ni; s->match_index[i] = 0; s->commit_index_sent[i] = 0;
  }
  s->status_mask = 0xFF;
  s->backoff_mask = 0;
  set_leader_id(s, s->self, fx);
  s->votes = 0;
  set_role(s, RGB_ROLE_LEADER, fx);
  fx->flags |= RGB_F_BECAME_LEADER;
}

/* one granted vote for a candidate in its own term, :1045-1061 */
static void candidate_vote_granted(oserver *sv, ofx *fx) {
  oscal *s = &sv->s;
  unsigned nv = (unsigned)s->votes + 1;
  if (nv == required_quorum(s)) become_leader(sv, fx);
  else s->votes = (uint8_t)nv;
}

static void vote_requests(oserver *sv, uint64_t term, int pre, ofx *fx) {
  uint64_t li, lt;
  log_last_index_term(&sv->log, &li, &lt);
  fx->flags &= ~(uint32_t)RGB_F_PRE_VOTE_REQS;   /* a single-member pre-vote goes straight on to the real vote */
  fx->flags |= RGB_F_SEND_VOTE_REQUESTS | (pre ? RGB_F_PRE_VOTE_REQS : 0);
  fx->has_reply = 0;
  fx->r_term = term; fx->r_next = pre ? sv->s.pre_vote_token : 0; fx->r_last = li; fx->r_lterm = lt;
  fx->vote_reqs = 1;
}

/* call_for_election(candidate, State) :2880-2899; the {next_event, cast, VoteForSelf} it emits
 * is the next message the reference processes, so it is applied here */
static void call_for_election_candidate(oserver *sv, ofx *fx) {
  oscal *s = &sv->s;
  uint64_t new_term = s->current_term + 1;
  update_term_and_voted_for(s, new_term, s->self, fx);
  set_leader_id(s, RGB_NONE, fx);
  s->votes = 0;
  set_role(s, RGB_ROLE_CANDIDATE, fx);
  vote_requests(sv, new_term, 0, fx);
  candidate_vote_granted(sv, fx);                         /* vote for self */
}

/* call_for_election(pre_vote, State) :2900-2924 (+ the self pre-vote, :1229-1246) */
static void call_for_election_pre_vote(oserver *sv, uint64_t token, ofx *fx) {
  oscal *s = &sv->s;
  update_term_and_voted_for(s, s->current_term, s->self, fx);
  set_leader_id(s, RGB_NONE, fx);
  s->votes = 0;
  s->pre_vote_token = token;
  set_role(s, RGB_ROLE_PRE_VOTE, fx);
  vote_requests(sv, s->current_term, 1, fx);
  /* self #pre_vote_result{vote_granted = true}: only counted when membership is voter */
  if (!s->self_nonvoter) {
    unsigned nv = (unsigned)s->votes + 1;
    if (nv == required_quorum(s)) call_for_election_candidate(sv, fx);
    else s->votes = (uint8_t)nv;
  }
}

static void pre_vote_reply(uint64_t term, uint64_t token, int granted, uint8_t to, ofx *fx) {
  fx->has_reply = 1;
  fx->flags |= RGB_F_REPLY | RGB_F_REPLY_PRE_VOTE | (granted ? RGB_F_REPLY_SUCCESS : 0);
  fx->r_term = term; fx->r_next = token; fx->r_last = 0; fx->r_lterm = 0;
  fx->reply_to = to;
}

/* process_pre_vote/3 :2926-2983 (the server stays in FsmState) */
static int process_pre_vote(oserver *sv, const rgb_msg *m, ofx *fx) {
  oscal *s = &sv->s;
  uint64_t token = m->c;
  if (m->term >= s->current_term) {
    update_term(s, m->term, fx);              /* pre-vote never sets voted_for */
    uint64_t li, lt;
    log_last_index_term(&sv->log, &li, &lt);
    int up = (m->b > lt) || (m->b == lt && m->a >= li);
    uint32_t theirs = m->n_entries, eff = s->effective_machine_version, ours = s->machine_version;
    if (up && (uint32_t)m->gap > RGB_PROTO_VERSION) {
      pre_vote_reply(m->term, token, 0, m->from, fx);
    } else if (up && (theirs == eff || (theirs >= eff && theirs <= ours))) {
      pre_vote_reply(m->term, token, 1, m->from, fx);
    } else if (up) {
      pre_vote_reply(m->term, token, 0, m->from, fx);
      fx->flags |= RGB_F_START_ELECTION_TIMEOUT;
    } else if (s->role == RGB_ROLE_FOLLOWER) {
      fx->flags |= RGB_F_START_ELECTION_TIMEOUT;          /* no reply, :2968-2969 */
    } else {
      pre_vote_reply(m->term, token, 0, m->from, fx);
    }
    return 0;
  }
  pre_vote_reply(s->current_term, token, 0, m->from, fx);
  return 0;
}

/* make_all_rpcs/1 :2353-2367 -> make_rpcs_for/2 :2369-2377: one append_entries_rpc (batch size 1)
 * to every peer with status normal; next_index is NOT advanced.  (heartbeat effects and
 * snapshot_backoff peers are outside the device model.) */
static int make_rpcs_for_peers(oserver *sv, uint32_t srv_id, int only_stale, ofx *fx) {
  oscal *s = &sv->s;
  update_heartbeat_rpc_effects(s, fx);                      /* :2349 / :2354-2355 */
  for (unsigned i = 0; i < s->n_members; i++) {
    if (i == s->self || !is_present(s, i)) continue;
    if (!((s->status_mask >> i) & 1u)) {
      /* make_all_rpcs/1 keeps {snapshot_backoff,_} peers and cancels their retry timers (:2356-2363);
       * stale_peers/1 (the tick) only takes normal ones */
      if (only_stale || !((s->backoff_mask >> i) & 1u)) continue;
      fx->cancel_mask |= (uint8_t)(1u << i);
      fx->flags |= RGB_F_CANCEL_SNAPSHOT_RETRY;
    }
    /* make_rpcs/1 on tick: stale_peers/1 :3012-3030 -- unconfirmed items or a newer commit index */
    if (only_stale && !(s->match_index[i] + 1 < s->next_index[i] || s->commit_index_sent[i] < s->commit_index))
      continue;
    uint64_t prev = s->next_index[i] - 1;
    uint64_t prev_term = log_fetch_term(&sv->log, prev);
    rgb_rpc r;
    memset(&r, 0, sizeof r);
    r.msg_index = fx->msg_index; r.server = srv_id; r.peer = (uint8_t)i;
    r.term = s->current_term; r.leader_commit = s->commit_index;
    if (prev_term == UNDEF && !(sv->log.snap_idx != UNDEF && sv->log.snap_idx == prev)) {
      if (sv->log.snap_idx == UNDEF || !(s->next_index[i] == 0 || prev < sv->log.snap_idx))
        return RGB_INV_PIPELINE_PREV_UNDEFINED;            /* PrevIdx = -1 is below any snapshot index */
      r.kind = RGB_RPC_SNAPSHOT; r.prev_log_index = sv->log.snap_idx; r.prev_log_term = sv->log.snap_term;
      r.next_index = sv->log.snap_idx;
      fx->flags |= RGB_F_SEND_SNAPSHOT;
    } else {
      if (prev_term == UNDEF) prev_term = sv->log.snap_term;
      uint64_t li, lt;
      log_last_index_term(&sv->log, &li, &lt);
      uint64_t to = prev + 1 < li ? prev + 1 : li;
      r.kind = RGB_RPC_AER; r.prev_log_index = prev; r.prev_log_term = prev_term;
      r.n_entries = (uint16_t)(to >= prev + 1 ? to - prev : 0);
      r.next_index = to + 1;
    }
    emit_rpc(fx, &r);
  }
  return 0;
}

static int make_all_rpcs(oserver *sv, uint32_t srv_id, ofx *fx) { return make_rpcs_for_peers(sv, srv_id, 0, fx); }

/* ------------------------------------------------------------- follower clauses -- */
static int follower_aer(oserver *sv, const rgb_msg *m, ofx *fx);

/* handle_follower(#append_entries_rpc{}), src/ra_server.erl:1283-1440 */
static int follower_aer(oserver *sv, const rgb_msg *m, ofx *fx) {
  oscal *s = &sv->s;
  olog *l = &sv->log;
  uint64_t cur_term = s->current_term;
  if (!(m->term >= cur_term)) {
    /* :1431-1440 lower term: reply CurTerm,false; state unchanged */
    aer_reply(sv, cur_term, 0, m->from, fx);
    return 0;
  }
  uint64_t pli = m->a, plt = m->b, leader_commit = m->c;
  uint64_t last_applied = s->last_applied;
  /* State0 = update_term(Term, State00#{leader_id => LeaderId}) :1298 */
  set_leader_id(s, m->from, fx);
  update_term(s, m->term, fx);
  int h = has_log_entry_or_snapshot(l, pli, plt);
  if (h == HLE_OK) {
    /* drop_existing/3 :3700-3708 -- one ra_log:exists per entry */
    uint32_t k = 0;
    uint64_t last_valid = pli;
    while (k < m->n_entries) {
      uint64_t idx = msg_first_index(m) + k;
      if (!log_exists(l, idx, msg_entry_term(m, k))) break;
      last_valid = idx;
      k++;
    }
    if (k == m->n_entries) {
      /* all entries already written, :1304-1364 */
      uint64_t local_last, lt;
      log_last_index_term(l, &local_last, &lt);
      int validated;
      if (m->n_entries == 0 && local_last > pli) {
        if (pli < last_applied) return RGB_INV_TRUNCATE_BELOW_APPLIED;
        int rc = log_set_last_index(l, pli);
        if (rc) return rc;
        fx->flags |= RGB_F_TRUNCATED;
        validated = 1;
      } else {
        validated = local_last <= last_valid;
      }
      if (validated) {
        s->commit_index = leader_commit;                   /* NOT clamped, :1331 */
        fx->flags |= RGB_F_LEADER_MSG;
        evaluate_commit_index_follower(sv, fx);
        aer_reply(sv, m->term, 1, m->from, fx);
      } else {
        /* :1344-1363: success reply up to what we had; term = PRE-update CurTerm;
         * the effect list holds only the cast (no record_leader_msg) */
        uint64_t v = last_applied > last_valid ? last_applied : last_valid;
        uint64_t vt = srv_fetch_term(l, v);
        fx->has_reply = 1;
        fx->flags |= RGB_F_REPLY | RGB_F_REPLY_SUCCESS;
        fx->r_term = cur_term; fx->r_next = v + 1; fx->r_last = v; fx->r_lterm = vt;
        fx->reply_to = m->from;
      }
      return 0;
    }
    /* new entries to write, :1365-1389 */
    uint64_t fst = msg_first_index(m) + k;
    s->commit_index = leader_commit;
    if (fst < last_applied) return RGB_INV_WRITE_BELOW_APPLIED;
    int rc = log_write(l, m, k);
    if (rc) return rc;
    fx->flags |= RGB_F_WROTE | RGB_F_LEADER_MSG;
    fx->w_first = fst; fx->w_last = msg_first_index(m) + (m->n_entries - 1);
    evaluate_commit_index_follower(sv, fx);
    return 0;                                              /* no reply until written */
  }
  if (h == HLE_MISSING) {
    /* :1390-1404 */
    aer_reply(sv, m->term, 0, m->from, fx);
    fx->flags |= RGB_F_LEADER_MSG;
    set_role(s, RGB_ROLE_AWAIT_CONDITION, fx);
    s->cond_reason = RGB_COND_MISSING;
  } else {
    /* term_mismatch :1405-1429, mismatch_append_entries_reply/3 :3614-3622 */
    uint64_t lat = srv_fetch_term(l, last_applied);
    if (lat == UNDEF) return RGB_INV_MISMATCH_TERM_UNDEFINED;
    fx->has_reply = 1;
    fx->flags |= RGB_F_REPLY | RGB_F_LEADER_MSG;
    fx->r_term = m->term; fx->r_next = last_applied + 1;
    fx->r_last = last_applied; fx->r_lterm = lat;
    fx->reply_to = m->from;
    set_role(s, RGB_ROLE_AWAIT_CONDITION, fx);
    s->cond_reason = RGB_COND_TERM_MISMATCH;
  }
  /* condition timeout repeats the reply effect, :1398-1403 / :1423-1428 */
  s->cond_reply[0] = fx->r_term; s->cond_reply[1] = fx->r_next;
  s->cond_reply[2] = fx->r_last; s->cond_reply[3] = fx->r_lterm;
  s->cond_leader = m->from;
  return 0;
}

/* handle_follower(#request_vote_rpc{}), src/ra_server.erl:1483-1529 */
static int follower_request_vote(oserver *sv, const rgb_msg *m, ofx *fx) {
  oscal *s = &sv->s;
  if (s->self_nonvoter) return 0;                          /* :1483-1488 ignored, no reply */
  uint8_t cand = m->from;
  if (m->term == s->current_term && s->voted_for != RGB_NONE && s->voted_for != cand) {
    vote_reply(m->term, 0, cand, fx);                      /* :1489-1497 */
    return 0;
  }
  if (m->term >= s->current_term) {
    update_term(s, m->term, fx);
    uint64_t li, lt;
    log_last_index_term(&sv->log, &li, &lt);
    /* is_candidate_log_up_to_date/3 :3157-3166 */
    int up = (m->b > lt) || (m->b == lt && m->a >= li);
    if (up) {
      update_term_and_voted_for(s, m->term, cand, fx);
      vote_reply(m->term, 1, cand, fx);
    } else {
      vote_reply(m->term, 0, cand, fx);
    }
    return 0;
  }
  vote_reply(s->current_term, 0, cand, fx);                /* :1522-1529 */
  return 0;
}

/* ra_log:handle_event({written,..}) in any role: *changed = last_written moved; the resend
 * request of a not_prefix written event is an effect flag (host I/O); returns an RGB_INV_* */
/* the RGB_MF_SEQX range list of the batch being stepped (ora_set_seq_ranges: test infrastructure, one list per
 * process; read-only while a step runs) */
static const uint64_t *g_seq_ranges = NULL;
static uint32_t g_n_seq_ranges = 0;
void ora_set_seq_ranges(ora_ctx *c, const uint64_t *ranges, uint32_t n_ranges) {
  (void)c;
  g_seq_ranges = n_ranges ? ranges : NULL; g_n_seq_ranges = n_ranges;
}

static int srv_written(oserver *sv, const rgb_msg *m, ofx *fx, int *changed) {
  int resend = 0, inv = 0;
  if (m->flags & RGB_MF_SEQX) {
    *changed = 0;
    if (!g_seq_ranges || m->n_entries == 0 || (uint64_t)m->c + m->n_entries > g_n_seq_ranges) return RGB_INV_WRITTEN_SEQ_LIST;
    *changed = log_written_sparse(&sv->log, m->term, g_seq_ranges + 2 * m->c, m->n_entries, m->run0_term, m->run1_term, 1,
                                  m->a, m->b, &resend, &inv);
  } else if (m->flags & RGB_MF_SEQ2)
    *changed = log_written_sparse(&sv->log, m->term, NULL, 0, m->run0_term, m->run1_term, 1, m->a, m->b, &resend, &inv);
  else
    *changed = log_written(&sv->log, m->term, m->a, m->b, &resend, &inv);
  if (inv) return inv;
  if (resend) fx->flags |= RGB_F_RESEND_PENDING;
  return 0;
}

/* handle_follower({ra_log_event,{written,..}}), src/ra_server.erl:1457-1474 */
static int follower_written(oserver *sv, const rgb_msg *m, ofx *fx) {
  int changed;
  int rc = srv_written(sv, m, fx, &changed);
  if (rc) return rc;
  if (changed && sv->s.leader_id != RGB_NONE)
    aer_reply(sv, sv->s.current_term, 1, sv->s.leader_id, fx);
  return 0;
}

/* handle_follower({ra_log_event, Evt}) for a snapshot_written event: same clause as written */
static int follower_snapshot_written(oserver *sv, const rgb_msg *m, ofx *fx) {
  int changed = log_snapshot_written(&sv->log, m->a, m->b);
  if (changed && sv->s.leader_id != RGB_NONE)
    aer_reply(sv, sv->s.current_term, 1, sv->s.leader_id, fx);
  return 0;
}

static int handle_follower(oserver *sv, const rgb_msg *m, ofx *fx) {
  switch (m->kind) {
    case RGB_MSG_SNAPSHOT_WRITTEN: return follower_snapshot_written(sv, m, fx);
    case RGB_MSG_AER:          return follower_aer(sv, m, fx);
    case RGB_MSG_REQUEST_VOTE: return follower_request_vote(sv, m, fx);
    case RGB_MSG_WRITTEN:      return follower_written(sv, m, fx);
    case RGB_MSG_AER_REPLY: {
      /* :1530-1533 Term = max(TheirTerm, CurTerm), update_term */
      update_term(&sv->s, m->term, fx);
      return 0;
    }
    case RGB_MSG_HEARTBEAT_RPC:
      if (m->term >= sv->s.current_term) {                 /* :1441-1450 */
        update_term(&sv->s, m->term, fx);
        set_leader_id(&sv->s, m->from, fx);
        heartbeat_reply(m->term, m->a, m->from, fx);
      } else {
        heartbeat_reply(sv->s.current_term, m->a, m->from, fx); /* :1451-1456 */
      }
      return 0;
    case RGB_MSG_HEARTBEAT_REPLY:                          /* :1534-1537 Term = max(TheirTerm, CurTerm) */
      update_term(&sv->s, m->term, fx);
      return 0;
    case RGB_MSG_VOTE_RESULT:  return 0;                   /* :1609-1611 ignored */
    case RGB_MSG_PRE_VOTE_RESULT: return 0;                /* :1612-1614 ignored */
    case RGB_MSG_PRE_VOTE_RPC:
      if (sv->s.self_nonvoter) return 0;                   /* :1475-1480 ignored, no reply */
      return process_pre_vote(sv, m, fx);                  /* :1481-1482 */
    case RGB_MSG_ELECTION_TIMEOUT:
      if (sv->s.self_nonvoter) return 0;                   /* :1619-1624 */
      call_for_election_pre_vote(sv, m->c, fx);            /* :1625-1626 */
      return 0;
    default:
      fx->flags |= RGB_F_UNHANDLED;                        /* :1655 catch-all */
      return 0;
  }
}

/* --------------------------------------------------------------- leader clauses -- */
static int handle_leader(struct ora_ctx *c, oserver *sv, uint32_t srv_id, const rgb_msg *m,
                         ofx *fx, int *reprocess) {
  oscal *s = &sv->s;
  olog *l = &sv->log;
  switch (m->kind) {
    case RGB_MSG_AER_REPLY: {
      unsigned peer = m->from;
      int success = (m->flags & RGB_MF_SUCCESS) != 0;
      if (success && m->term == s->current_term) {
        /* :532-571 */
        if (!is_present(s, peer)) return 0;
        if (m->b > s->match_index[peer]) s->match_index[peer] = m->b;
        if (m->a > s->next_index[peer]) s->next_index[peer] = m->a;
        evaluate_quorum(sv, fx);
        fx->flags |= RGB_F_PIPELINE;
        return 0;
      }
      if (m->term > s->current_term) {
        /* :572-586 */
        if (!is_present(s, peer)) return 0;
        set_leader_id(s, RGB_NONE, fx);
        update_term(s, m->term, fx);
        set_role(s, RGB_ROLE_FOLLOWER, fx);
        return 0;
      }
      if (!success) {
        /* :587-652 (no term guard) */
        if (!is_present(s, peer)) return 0;
        uint64_t mi = s->match_index[peer], ni = s->next_index[peer];
        uint64_t peer_next = m->a, peer_last = m->b, peer_last_term = m->c;
        uint64_t t = log_fetch_term(l, peer_last);         /* ra_log:fetch_term: NO snapshot fallback */
        if (t == UNDEF) {
          s->next_index[peer] = peer_next;
        } else if (t == peer_last_term && peer_last >= mi) {
          s->match_index[peer] = peer_last;
          s->next_index[peer] = peer_next;
        } else if (peer_last < mi) {
          s->match_index[peer] = peer_last;
          s->next_index[peer] = peer_last + 1;
        } else {
          /* NextIndex = max(min(NI-1, PeerLastIdx), MI + 1); NI-1 may be -1 in Erlang */
          int64_t a = (int64_t)ni - 1;
          int64_t b = (int64_t)peer_last;
          int64_t mn = a < b ? a : b;
          int64_t lo = (int64_t)mi + 1;
          s->next_index[peer] = (uint64_t)(mn > lo ? mn : lo);
        }
        int more;
        return make_pipelined_rpc_effects(c, sv, srv_id, 0, &more, fx);
      }
      fx->flags |= RGB_F_UNHANDLED;                        /* stale-term success reply: :1038-1040 */
      return 0;
    }
    case RGB_MSG_AER: {
      if (m->term > s->current_term) {
        /* :835-844 */
        set_leader_id(s, RGB_NONE, fx);
        update_term(s, m->term, fx);
        set_role(s, RGB_ROLE_FOLLOWER, fx);
        *reprocess = 1;
        return 0;
      }
      if (m->term == s->current_term) return RGB_INV_LEADER_SAW_AER_SAME_TERM;
      aer_reply(sv, s->current_term, 0, m->from, fx);      /* :850-854 */
      return 0;
    }
    case RGB_MSG_REQUEST_VOTE: {
      if (m->term > s->current_term) {
        /* :928-942 */
        if (!is_present(s, m->from)) return 0;
        set_leader_id(s, RGB_NONE, fx);
        update_term(s, m->term, fx);
        set_role(s, RGB_ROLE_FOLLOWER, fx);
        *reprocess = 1;
        return 0;
      }
      vote_reply(s->current_term, 0, m->from, fx);         /* :943-945 */
      return 0;
    }
    case RGB_MSG_WRITTEN: {
      /* :739-744 */
      int changed;
      int rc = srv_written(sv, m, fx, &changed);
      if (rc) return rc;
      evaluate_quorum(sv, fx);
      fx->flags |= RGB_F_PIPELINE;
      return 0;
    }
    case RGB_MSG_PIPELINE_RPCS: {
      if (m->flags & RGB_MF_TICK)                          /* leader tick_timeout: make_rpcs/1 :2348-2351 */
        return make_rpcs_for_peers(sv, srv_id, 1, fx);
      /* :793-801 */
      int more;
      int rc = make_pipelined_rpc_effects(c, sv, srv_id, 0, &more, fx);
      if (rc) return rc;
      if (more) fx->flags |= RGB_F_PIPELINE;
      return 0;
    }
    case RGB_MSG_APPEND: {
      /* {command,_} :653-693 / {commands,_} :695-738: ra_log:append per command at
       * next_index in the current term, then make_pipelined_rpc_effects */
      for (uint32_t k = 0; k < m->n_entries; k++)
        if (log_append(l, log_next_index(l), s->current_term)) return RGB_INV_WRITE_INTEGRITY;
      int more;
      return make_pipelined_rpc_effects(c, sv, srv_id, (m->flags & RGB_MF_FORCE) != 0, &more, fx);
    }
    case RGB_MSG_PRE_VOTE_RPC: {
      if (m->term > s->current_term) {
        /* :946-960 */
        if (!is_present(s, m->from)) return 0;
        set_leader_id(s, RGB_NONE, fx);
        update_term(s, m->term, fx);
        set_role(s, RGB_ROLE_FOLLOWER, fx);
        *reprocess = 1;
        return 0;
      }
      return make_all_rpcs(sv, srv_id, fx);                /* :961-966 enforce leadership */
    }
    case RGB_MSG_VOTE_RESULT:                              /* :967-969 */
    case RGB_MSG_PRE_VOTE_RESULT:                          /* :970-972 */
      return 0;
    case RGB_MSG_CONSISTENT_QUERY:
      /* :855-860 / :868-873 with cluster_change_permitted = true (the host holds queries while it
       * is false, :861-867) */
      make_heartbeat_rpc_effects(s, fx);
      return 0;
    case RGB_MSG_HEARTBEAT_RPC:
      if (m->term > s->current_term) {                     /* :880-889 */
        set_leader_id(s, RGB_NONE, fx);
        update_term(s, m->term, fx);
        set_role(s, RGB_ROLE_FOLLOWER, fx);
        *reprocess = 1;
        return 0;
      }
      if (s->current_term > m->term) {                     /* :890-897 */
        heartbeat_reply(s->current_term, m->a, m->from, fx);
        return 0;
      }
      return RGB_INV_LEADER_SAW_HEARTBEAT_SAME_TERM;       /* :898-903 */
    case RGB_MSG_HEARTBEAT_REPLY:                          /* :904-927 */
      if (m->term == s->current_term) {
        heartbeat_rpc_quorum(s, m->a, m->from, fx);
      } else if (m->term > s->current_term) {
        set_leader_id(s, RGB_NONE, fx);
        update_term(s, m->term, fx);
        set_role(s, RGB_ROLE_FOLLOWER, fx);
      }
      return 0;
    case RGB_MSG_SNAPSHOT_WRITTEN:                         /* :745-747 other ra_log_events */
      log_snapshot_written(l, m->a, m->b);
      return 0;
    default:
      fx->flags |= RGB_F_UNHANDLED;
      return 0;
  }
}

/* ------------------------------------------------------------ candidate clauses -- */
static int handle_candidate(oserver *sv, const rgb_msg *m, ofx *fx, int *reprocess) {
  oscal *s = &sv->s;
  switch (m->kind) {
    case RGB_MSG_VOTE_RESULT: {
      int granted = (m->flags & RGB_MF_SUCCESS) != 0;
      if (granted && m->term == s->current_term) {
        candidate_vote_granted(sv, fx);                     /* :1045-1061 */
        return 0;
      }
      if (m->term > s->current_term) {
        update_term_and_voted_for(s, m->term, RGB_NONE, fx); /* :1062-1069 */
        set_role(s, RGB_ROLE_FOLLOWER, fx);
        return 0;
      }
      return 0;                                             /* :1070-1071, :1131-1133 */
    }
    case RGB_MSG_AER: {
      if (m->term >= s->current_term) {
        update_term_and_voted_for(s, m->term, RGB_NONE, fx); /* :1072-1075 */
        set_role(s, RGB_ROLE_FOLLOWER, fx);
        *reprocess = 1;
        return 0;
      }
      aer_reply(sv, s->current_term, 0, m->from, fx);       /* :1076-1080 */
      return 0;
    }
    case RGB_MSG_AER_REPLY: {
      if (m->term > s->current_term) {
        update_term_and_voted_for(s, m->term, RGB_NONE, fx); /* :1098-1106 */
        set_role(s, RGB_ROLE_FOLLOWER, fx);
        return 0;
      }
      fx->flags |= RGB_F_UNHANDLED;
      return 0;
    }
    case RGB_MSG_REQUEST_VOTE: {
      if (m->term > s->current_term) {
        update_term_and_voted_for(s, m->term, RGB_NONE, fx); /* :1107-1114 */
        set_role(s, RGB_ROLE_FOLLOWER, fx);
        *reprocess = 1;
        return 0;
      }
      vote_reply(s->current_term, 0, m->from, fx);          /* :1123-1125 */
      return 0;
    }
    case RGB_MSG_WRITTEN:
      { int changed; return srv_written(sv, m, fx, &changed); }   /* :1157-1160 */
    case RGB_MSG_PRE_VOTE_RPC:
      if (m->term > s->current_term) {
        update_term_and_voted_for(s, m->term, RGB_NONE, fx); /* :1116-1122 */
        set_role(s, RGB_ROLE_FOLLOWER, fx);
        *reprocess = 1;
        return 0;
      }
      return process_pre_vote(sv, m, fx);                   /* :1127-1131 */
    case RGB_MSG_HEARTBEAT_RPC:
      if (m->term >= s->current_term) {                     /* :1081-1084 */
        update_term_and_voted_for(s, m->term, RGB_NONE, fx);
        set_role(s, RGB_ROLE_FOLLOWER, fx);
        *reprocess = 1;
        return 0;
      }
      heartbeat_reply(s->current_term, m->a, m->from, fx);  /* :1085-1090 */
      return 0;
    case RGB_MSG_HEARTBEAT_REPLY:
      if (m->term > s->current_term) {                      /* :1091-1099 */
        update_term_and_voted_for(s, m->term, RGB_NONE, fx);
        set_role(s, RGB_ROLE_FOLLOWER, fx);
        return 0;
      }
      fx->flags |= RGB_F_UNHANDLED;                         /* catch-all: {error, unsupported_call} */
      return 0;
    case RGB_MSG_PRE_VOTE_RESULT: return 0;                 /* :1135-1137 */
    case RGB_MSG_SNAPSHOT_WRITTEN:
      log_snapshot_written(&sv->log, m->a, m->b);           /* :1157-1160 */
      return 0;
    case RGB_MSG_ELECTION_TIMEOUT:
      call_for_election_candidate(sv, fx);                  /* :1161-1162 */
      return 0;
    default:
      fx->flags |= RGB_F_UNHANDLED;
      return 0;
  }
}

/* ------------------------------------------------------------- pre_vote clauses -- */
static int handle_pre_vote(oserver *sv, const rgb_msg *m, ofx *fx, int *reprocess) {
  oscal *s = &sv->s;
  switch (m->kind) {
    case RGB_MSG_AER:
      if (m->term >= s->current_term) {
        update_term(s, m->term, fx);                        /* :1192-1197 */
        s->votes = 0;
        set_role(s, RGB_ROLE_FOLLOWER, fx);
        *reprocess = 1;
        return 0;
      }
      fx->flags |= RGB_F_UNHANDLED;
      return 0;
    case RGB_MSG_REQUEST_VOTE:
      if (m->term > s->current_term) {
        update_term(s, m->term, fx);                        /* :1214-1219 */
        s->votes = 0;
        set_role(s, RGB_ROLE_FOLLOWER, fx);
        *reprocess = 1;
        return 0;
      }
      fx->flags |= RGB_F_UNHANDLED;
      return 0;
    case RGB_MSG_VOTE_RESULT:
      return 0;                                             /* :1249-1251 */
    case RGB_MSG_WRITTEN:
      { int changed; return srv_written(sv, m, fx, &changed); }   /* :1257-1260 */
    case RGB_MSG_PRE_VOTE_RESULT: {
      int granted = (m->flags & RGB_MF_SUCCESS) != 0;
      if (m->term > s->current_term) {
        update_term(s, m->term, fx);                        /* :1219-1228 */
        s->votes = 0;
        set_role(s, RGB_ROLE_FOLLOWER, fx);
        return 0;
      }
      if (granted && m->term == s->current_term && m->c == s->pre_vote_token && !s->self_nonvoter) {
        /* :1229-1246 */
        unsigned nv = (unsigned)s->votes + 1;
        if (nv == required_quorum(s)) call_for_election_candidate(sv, fx);
        else s->votes = (uint8_t)nv;
        return 0;
      }
      return 0;                                             /* :1247-1249 */
    }
    case RGB_MSG_HEARTBEAT_RPC:
      if (m->term >= s->current_term) {                     /* :1198-1203 */
        update_term(s, m->term, fx);
        s->votes = 0;
        set_role(s, RGB_ROLE_FOLLOWER, fx);
        *reprocess = 1;
        return 0;
      }
      heartbeat_reply(s->current_term, m->a, m->from, fx);  /* :1204-1208 */
      return 0;
    case RGB_MSG_HEARTBEAT_REPLY:
      if (m->term > s->current_term) {                      /* :1209-1212 */
        update_term(s, m->term, fx);
        s->votes = 0;
        set_role(s, RGB_ROLE_FOLLOWER, fx);
        return 0;
      }
      fx->flags |= RGB_F_UNHANDLED;                         /* catch-all */
      return 0;
    case RGB_MSG_SNAPSHOT_WRITTEN:
      log_snapshot_written(&sv->log, m->a, m->b);           /* :1257-1260 */
      return 0;
    case RGB_MSG_PRE_VOTE_RPC:
      return process_pre_vote(sv, m, fx);                   /* :1250-1251 */
    case RGB_MSG_ELECTION_TIMEOUT:
      call_for_election_pre_vote(sv, m->c, fx);             /* :1255-1256 */
      return 0;
    default:
      fx->flags |= RGB_F_UNHANDLED;
      return 0;
  }
}

/* ------------------------------------------------------ await_condition clauses -- */
static int handle_await_condition(oserver *sv, const rgb_msg *m, ofx *fx, int *reprocess) {
  oscal *s = &sv->s;
  /* condition = #{predicate_fun => fun wal_down_condition/2} (src/ra_server.erl:1377-1385, 2232-2233): the predicate is
   * ra_log:can_write(Log), host knowledge carried by RGB_MF_CAN_WRITE.  The follower's condition map has no
   * transition_to and no timeout (defaults: follower, no effects); the leader's (:660-668) has transition_to => leader
   * and timeout => #{effects => [{next_event, cast, {transfer_leadership, PeerId}}] (none without a peer),
   * transition_to => leader} */
  if (s->cond_reason == RGB_COND_WAL_DOWN || s->cond_reason == RGB_COND_WAL_DOWN_LEADER) {
    const int to_leader = s->cond_reason == RGB_COND_WAL_DOWN_LEADER;
    const uint8_t back = to_leader ? RGB_ROLE_LEADER : RGB_ROLE_FOLLOWER;
    switch (m->kind) {
      case RGB_MSG_REQUEST_VOTE: case RGB_MSG_PRE_VOTE_RPC: case RGB_MSG_ELECTION_TIMEOUT:
      case RGB_MSG_WRITTEN: case RGB_MSG_SNAPSHOT_WRITTEN:
        break;                                               /* their own clauses, below */
      case RGB_MSG_AWAIT_TIMEOUT:
        /* :1932-1945: predicate true -> transition_to, no effects; false -> the timeout's transition_to and effects */
        if (to_leader && !(m->flags & RGB_MF_CAN_WRITE) &&
            (s->present_mask & ~(1u << s->self)) != 0)       /* maps:to_list(maps:remove(Self, Cluster)) =/= [] */
          fx->flags |= RGB_F_TRANSFER_LEADERSHIP;
        set_role(s, back, fx);
        return 0;
      default:
        if (m->flags & RGB_MF_CAN_WRITE) {                   /* :1950-1955 {next_event, Msg} */
          set_role(s, back, fx);
          *reprocess = 1;
        }
        return 0;
    }
  }
  switch (m->kind) {
    case RGB_MSG_REQUEST_VOTE:
      set_role(s, RGB_ROLE_FOLLOWER, fx);                   /* :1918-1919 */
      *reprocess = 1;
      return 0;
    case RGB_MSG_AWAIT_TIMEOUT: {
      /* :1932-1945; follower_catchup_cond(_, _Msg, _) -> false :2229-2230: replay the stored
       * effects [cast reply, record_leader_msg] and return to follower */
      fx->has_reply = 1;
      fx->flags |= RGB_F_REPLY | RGB_F_LEADER_MSG;
      fx->r_term = s->cond_reply[0]; fx->r_next = s->cond_reply[1];
      fx->r_last = s->cond_reply[2]; fx->r_lterm = s->cond_reply[3];
      fx->reply_to = s->cond_leader;
      set_role(s, RGB_ROLE_FOLLOWER, fx);
      return 0;
    }
    case RGB_MSG_WRITTEN:
      { int changed; return srv_written(sv, m, fx, &changed); }   /* :1946-1949, no reply */
    case RGB_MSG_SNAPSHOT_WRITTEN:
      log_snapshot_written(&sv->log, m->a, m->b);           /* :1946-1949 */
      return 0;
    case RGB_MSG_PRE_VOTE_RPC:
      return process_pre_vote(sv, m, fx);                   /* :1920-1921 */
    case RGB_MSG_ELECTION_TIMEOUT:
      if (s->self_nonvoter) return 0;                       /* :1922-1929 */
      call_for_election_pre_vote(sv, m->c, fx);             /* :1930-1931 */
      return 0;
    case RGB_MSG_AER: {
      /* follower_catchup_cond/3 :2201-2218 */
      int pred = 0;
      if (m->term >= s->current_term) {
        int h = has_log_entry_or_snapshot(&sv->log, m->a, m->b);
        if (h == HLE_OK) pred = 1;
        else if (h == HLE_MISMATCH) pred = (s->cond_reason == RGB_COND_MISSING);
      }
      if (pred) {
        set_role(s, RGB_ROLE_FOLLOWER, fx);                 /* :1950-1955 {next_event, Msg} */
        *reprocess = 1;
      }
      return 0;                                             /* false: no effects, state kept */
    }
    default:
      return 0;                                             /* predicate false: stay, no effects */
  }
}

/* ------------------------------------------------------------------ dispatcher --- */
static void process_one(struct ora_ctx *c, uint32_t msg_index, const rgb_msg *m,
                        rgb_decision *d, rgb_rpc *rpcs, uint32_t rpc_cap, uint32_t *n_rpcs) {
  memset(d, 0, sizeof *d);
  d->server = m->server;
  d->kind = m->kind;
  d->reply_to = RGB_NONE;
  if (m->kind == RGB_MSG_NOP) return;
  if (m->server >= c->n_servers) { d->flags = RGB_F_UNHANDLED; d->role = 0xFF; return; }
  oserver *sv = &c->sv[m->server];
  oscal saved = sv->s;                                      /* a crash leaves the old state */
  sv->log.pend_orig = sv->log.pend_idx;
  olog saved_log = sv->log;                                 /* cursors only: ra_log:append (the one
                                                               edit a later assertion can follow)
                                                               writes above the old last index */
  ofx fx;
  memset(&fx, 0, sizeof fx);
  fx.reply_to = RGB_NONE;
  fx.rpcs = rpcs; fx.rpc_cap = rpc_cap; fx.n_rpcs_total = *n_rpcs; fx.msg_index = msg_index;
  uint32_t rpcs_before = *n_rpcs;
  int rc = 0;
  for (int pass = 0; pass < 3; pass++) {                    /* await_condition -> leader -> follower at most */
    int reprocess = 0;
    switch (sv->s.role) {
      case RGB_ROLE_FOLLOWER:        rc = handle_follower(sv, m, &fx); break;
      case RGB_ROLE_LEADER:          rc = handle_leader(c, sv, m->server, m, &fx, &reprocess); break;
      case RGB_ROLE_CANDIDATE:       rc = handle_candidate(sv, m, &fx, &reprocess); break;
      case RGB_ROLE_PRE_VOTE:        rc = handle_pre_vote(sv, m, &fx, &reprocess); break;
      case RGB_ROLE_AWAIT_CONDITION: rc = handle_await_condition(sv, m, &fx, &reprocess); break;
      default: fx.flags |= RGB_F_UNHANDLED; break;
    }
    if (rc || !reprocess) break;
    fx.flags |= RGB_F_REPROCESSED;                          /* {next_event, Msg}: runs next */
  }
  if (rc) {
    sv->s = saved;
    saved_log.terms = sv->log.terms; saved_log.cap = sv->log.cap; saved_log.base = sv->log.base;
    if (sv->log.pend_idx && sv->log.pend_idx != saved_log.pend_idx) free(sv->log.pend_idx);
    sv->log = saved_log;
    d->role = saved.role;
    d->flags = RGB_F_INVARIANT;
    d->invariant = (uint16_t)rc;
    d->commit_index = saved.commit_index;
    d->last_applied = saved.last_applied;
    *n_rpcs = rpcs_before;
    return;
  }
  /* The engine keeps the log's term structure as at most max_runs (first index, term) runs; when a
   * message leaves more, it forgets the oldest runs -- the range then starts at the oldest run it
   * still knows -- and raises RGB_F_RUNS_OVERFLOW (include/ra_gpu_batch.h).  Not reference
   * behaviour: a bound of the engine, modelled here so that parity also covers it. */
  if (c->max_runs && sv->log.has_range) {
    olog *l = &sv->log;
    uint32_t runs = 1;
    for (uint64_t i = l->first + 1; i <= l->last; i++)
      if (l->terms[i - l->base] != l->terms[i - 1 - l->base]) runs++;
    if (runs > c->max_runs) {
      uint32_t drop = runs - c->max_runs;
      uint64_t i = l->first + 1;
      for (; i <= l->last && drop; i++)
        if (l->terms[i - l->base] != l->terms[i - 1 - l->base]) drop--;
      l->first = i - 1;                                     /* start of the oldest run kept */
      fx.flags |= RGB_F_RUNS_OVERFLOW;
    }
  }
  if (saved_log.pend_idx && saved_log.pend_idx != sv->log.pend_idx) free(saved_log.pend_idx);
  sv->log.pend_orig = sv->log.pend_idx;
  *n_rpcs = fx.n_rpcs_total;
  d->role = sv->s.role;
  d->flags = fx.flags;
  d->n_rpcs = fx.n_rpcs;
  if (fx.has_reply || fx.vote_reqs) {
    d->reply_to = fx.has_reply ? fx.reply_to : RGB_NONE;
    d->reply_term = fx.r_term; d->reply_next_index = fx.r_next;
    d->reply_last_index = fx.r_last; d->reply_last_term = fx.r_lterm;
  } else if (fx.flags & RGB_F_WROTE) {
    d->reply_next_index = fx.w_first; d->reply_last_index = fx.w_last;
  }
  if ((fx.flags & RGB_F_SEND_HEARTBEATS) || m->kind == RGB_MSG_CONSISTENT_QUERY) {
    /* never together with a reply: #heartbeat_rpc{term, query_index} for the peers in heartbeat_to;
     * for a consistent query also the index the host queues the query under */
    d->heartbeat_to = fx.hb_mask;
    d->reply_term = fx.hb_term; d->reply_last_term = fx.hb_query_index;
  }
  if (fx.flags & RGB_F_QUERY_QUORUM) d->reply_next_index = fx.q_consensus;
  d->cancel_backoff = fx.cancel_mask;
  d->commit_index = sv->s.commit_index;
  d->last_applied = sv->s.last_applied;
}

/* -------------------------------------------------------------------- public ----- */
static void server_init_empty(oserver *sv, uint32_t n_members, uint32_t self) {
  /* ra_server:init/1 on an empty log == empty_state/2 of test/ra_server_SUITE.erl:4139-4149:
   * term 0, log [0:0] written, peers new_peer/0 (src/ra_server.erl:2990-2995) */
  memset(&sv->s, 0, sizeof sv->s);
  sv->s.role = RGB_ROLE_FOLLOWER;
  sv->s.self = (uint8_t)self;
  sv->s.n_members = (uint8_t)n_members;
  sv->s.voted_for = RGB_NONE; sv->s.leader_id = RGB_NONE; sv->s.cond_leader = RGB_NONE;
  sv->s.present_mask = (uint8_t)((1u << n_members) - 1u);
  sv->s.voter_mask = sv->s.present_mask;
  sv->s.status_mask = 0xFF;
  for (unsigned i = 0; i < n_members; i++) sv->s.next_index[i] = 1;
  olog *l = &sv->log;
  free(l->terms);
  free(l->pend_idx);
  memset(l, 0, sizeof *l);
  l->snap_idx = UNDEF; l->snap_term = UNDEF;
  log_append(l, 0, 0);                                      /* src/ra_log.erl:1637-1647 */
  l->lw_idx = 0; l->lw_term = 0;
  l->pend_first = 1;                                        /* ...and its written event: nothing pending */
}

ora_ctx *ora_new(uint32_t n_groups, uint32_t n_members, uint32_t max_pipeline_count,
                 uint32_t max_aer_batch) {
  if (n_members == 0 || n_members > RGB_MAX_MEMBERS) return NULL;
  ora_ctx *c = (ora_ctx *)calloc(1, sizeof *c);
  if (!c) return NULL;
  c->n_members = n_members;
  c->n_servers = n_groups * n_members;
  c->max_pipeline_count = max_pipeline_count ? max_pipeline_count : RGB_DEFAULT_MAX_PIPELINE_COUNT;
  c->max_aer_batch = max_aer_batch ? max_aer_batch : RGB_AER_CHUNK_SIZE;
  c->sv = (oserver *)calloc(c->n_servers ? c->n_servers : 1, sizeof(oserver));
  if (!c->sv) { free(c); return NULL; }
  for (uint32_t i = 0; i < c->n_servers; i++) server_init_empty(&c->sv[i], n_members, i % n_members);
  return c;
}

void ora_free(ora_ctx *c) {
  if (!c) return;
  for (uint32_t i = 0; i < c->n_servers; i++) { free(c->sv[i].log.terms); free(c->sv[i].log.pend_idx); }
  free(c->sv);
  free(c);
}

uint32_t ora_n_servers(const ora_ctx *c) { return c->n_servers; }

int ora_set_state(ora_ctx *c, uint32_t first, uint32_t n, const rgb_server_state *in) {
  if ((uint64_t)first + n > c->n_servers) return RGB_E_INVAL;
  for (uint32_t k = 0; k < n; k++) {
    const rgb_server_state *h = &in[k];
    oserver *sv = &c->sv[first + k];
    oscal *s = &sv->s;
    if (h->n_runs > RGB_MAX_RUNS || h->n_members > RGB_MAX_MEMBERS) return RGB_E_INVAL;
    s->current_term = h->current_term; s->commit_index = h->commit_index;
    s->last_applied = h->last_applied;
    memcpy(s->cond_reply, h->cond_reply, sizeof s->cond_reply);
    memcpy(s->match_index, h->match_index, sizeof s->match_index);
    memcpy(s->next_index, h->next_index, sizeof s->next_index);
    memcpy(s->commit_index_sent, h->commit_index_sent, sizeof s->commit_index_sent);
    s->role = h->role; s->cond_reason = h->cond_reason; s->self = h->self;
    s->n_members = h->n_members; s->voted_for = h->voted_for; s->leader_id = h->leader_id;
    s->votes = h->votes; s->present_mask = h->present_mask; s->voter_mask = h->voter_mask;
    s->status_mask = h->status_mask; s->self_nonvoter = h->self_nonvoter;
    /* canonical: backed-off peers are members, not self, not normal */
    s->backoff_mask = (uint8_t)(h->backoff_mask & ~h->status_mask & h->present_mask & ~(1u << h->self));
    s->cond_leader = h->cond_leader;
    s->pre_vote_token = h->pre_vote_token;
    s->machine_version = h->machine_version;
    s->effective_machine_version = h->effective_machine_version;
    s->query_index = h->query_index;
    memcpy(s->peer_query_index, h->peer_query_index, sizeof s->peer_query_index);
    olog *l = &sv->log;
    free(l->terms);
    free(l->pend_idx);
    memset(l, 0, sizeof *l);
    l->snap_idx = h->snapshot_index; l->snap_term = h->snapshot_term;
    l->lw_idx = h->last_written_index; l->lw_term = h->last_written_term;
    l->last_term = h->last_term;
    l->pend_first = h->pending_first;
    if (h->first_index <= h->last_index) {
      if (h->n_runs == 0 || h->run_start[0] != h->first_index) return RGB_E_INVAL;
      l->has_range = 1; l->first = h->first_index; l->last = h->last_index;
      if (log_reserve(l, l->first) || log_reserve(l, l->last)) return RGB_E_NOMEM;
      for (unsigned r = 0; r < h->n_runs; r++) {
        uint64_t a = h->run_start[r];
        uint64_t b = (r + 1 < h->n_runs) ? h->run_start[r + 1] - 1 : h->last_index;
        if (b < a || b > h->last_index) return RGB_E_INVAL;
        for (uint64_t i = a; i <= b; i++) l->terms[i - l->base] = h->run_term[r];
      }
    } else {
      l->has_range = 0; l->first = h->first_index; l->last = h->last_index;
    }
    if (h->n_pending_old) {
      /* sparse pending: the old ranges and the newest range as one explicit index list */
      if (h->n_pending_old > 2) return RGB_E_INVAL;
      size_t n = 0;
      for (unsigned k = 0; k < h->n_pending_old; k++) {
        if (h->pending_old[k][0] > h->pending_old[k][1]) return RGB_E_INVAL;
        n += (size_t)(h->pending_old[k][1] - h->pending_old[k][0] + 1);
      }
      const int newest = l->has_range && h->pending_first <= h->last_index;
      if (newest) n += (size_t)(h->last_index - h->pending_first + 1);
      uint64_t *v = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
      if (!v) return RGB_E_NOMEM;
      size_t j = 0;
      for (unsigned k = 0; k < h->n_pending_old; k++)
        for (uint64_t i = h->pending_old[k][0]; i <= h->pending_old[k][1]; i++) v[j++] = i;
      if (newest) for (uint64_t i = h->pending_first; i <= h->last_index; i++) v[j++] = i;
      for (size_t q = 1; q < j; q++) if (v[q] <= v[q - 1]) { free(v); return RGB_E_INVAL; }
      l->pend_idx = v; l->pend_n = j; l->pend_orig = NULL;
      pend_sync(l);
      l->pend_orig = l->pend_idx;
    }
  }
  return RGB_OK;
}

int ora_get_state(const ora_ctx *c, uint32_t first, uint32_t n, rgb_server_state *out) {
  if ((uint64_t)first + n > c->n_servers) return RGB_E_INVAL;
  for (uint32_t k = 0; k < n; k++) {
    rgb_server_state *h = &out[k];
    const oserver *sv = &c->sv[first + k];
    const oscal *s = &sv->s;
    const olog *l = &sv->log;
    memset(h, 0, sizeof *h);
    h->current_term = s->current_term; h->commit_index = s->commit_index;
    h->last_applied = s->last_applied;
    log_last_index_term(l, &h->last_index, &h->last_term);
    h->last_written_index = l->lw_idx; h->last_written_term = l->lw_term;
    h->pending_first = l->pend_first;
    if (l->pend_idx) {
      /* the runs of the list below the newest range [pend_first .. last] */
      size_t top = l->pend_n;
      while (top > 0 && l->pend_idx[top - 1] >= l->pend_first) top--;
      unsigned nr = 0;
      for (size_t i = 0; i < top; i++) {
        if (i == 0 || l->pend_idx[i] != l->pend_idx[i - 1] + 1) {
          if (nr == 2) { nr = 3; break; }                  /* more ranges than the boundary holds */
          h->pending_old[nr][0] = l->pend_idx[i];
          nr++;
        }
        h->pending_old[nr - 1][1] = l->pend_idx[i];
      }
      h->n_pending_old = (uint8_t)nr;
    }
    h->snapshot_index = l->snap_idx; h->snapshot_term = l->snap_term;
    memcpy(h->cond_reply, s->cond_reply, sizeof s->cond_reply);
    memcpy(h->match_index, s->match_index, sizeof s->match_index);
    memcpy(h->next_index, s->next_index, sizeof s->next_index);
    memcpy(h->commit_index_sent, s->commit_index_sent, sizeof s->commit_index_sent);
    h->role = s->role; h->cond_reason = s->cond_reason; h->self = s->self;
    h->n_members = s->n_members; h->voted_for = s->voted_for; h->leader_id = s->leader_id;
    h->votes = s->votes; h->present_mask = s->present_mask; h->voter_mask = s->voter_mask;
    h->status_mask = s->status_mask; h->self_nonvoter = s->self_nonvoter;
    h->backoff_mask = s->backoff_mask;
    h->cond_leader = s->cond_leader;
    h->query_index = s->query_index;
    memcpy(h->peer_query_index, s->peer_query_index, sizeof s->peer_query_index);
    h->pre_vote_token = s->pre_vote_token;
    h->machine_version = s->machine_version;
    h->effective_machine_version = s->effective_machine_version;
    if (l->has_range) {
      h->first_index = l->first;
      unsigned nr = 0;
      int overflow = 0;
      for (uint64_t i = l->first; i <= l->last; i++) {
        uint64_t t = l->terms[i - l->base];
        if (nr == 0 || h->run_term[nr - 1] != t) {
          if (nr == RGB_MAX_RUNS) { overflow = 1; break; }
          h->run_start[nr] = i; h->run_term[nr] = t; nr++;
        }
      }
      h->n_runs = (uint8_t)nr;
      if (overflow) return RGB_E_UNSUPPORTED;
    } else {
      h->first_index = h->last_index + 1;
      h->n_runs = 0;
    }
  }
  return RGB_OK;
}

int ora_step(ora_ctx *c, const rgb_msg *msgs, uint32_t n, rgb_decision *out,
             rgb_rpc *rpcs, uint32_t rpc_cap, uint32_t *n_rpcs) {
  uint32_t nr = 0;
  for (uint32_t i = 0; i < n; i++)
    process_one(c, i, &msgs[i], &out[i], rpcs, rpc_cap, &nr);
  if (n_rpcs) *n_rpcs = nr;
  return RGB_OK;
}

/* Parallel form used only for the CPU baseline: a tick holds at most one message per server,
 * so servers are independent and messages can be processed by any thread.  rpc records are
 * discarded (counted only).  Compiled with -fopenmp; n_threads <= 0 keeps the default. */
#ifdef _OPENMP
#include <omp.h>
#endif
int ora_step_parallel(ora_ctx *c, const rgb_msg *msgs, uint32_t n, rgb_decision *out,
                      int n_threads, uint64_t *n_rpcs_total) {
  uint64_t total = 0;
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(static) reduction(+ : total)
#endif
  for (int64_t i = 0; i < (int64_t)n; i++) {
    uint32_t nr = 0;
    process_one(c, (uint32_t)i, &msgs[i], &out[i], NULL, 0, &nr);
    total += nr;
  }
  if (n_rpcs_total) *n_rpcs_total = total;
  (void)n_threads;
  return RGB_OK;
}

void ora_set_max_runs(ora_ctx *c, uint32_t max_runs) { if (c) c->max_runs = max_runs; }

int ora_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* the same canonical-state checksum the device computes (rgb_state_checksum): FNV-1a over the
 * words of the canonical form; lets full-size runs be compared without downloading state. */
static uint64_t fnv_word(uint64_t h, uint64_t w) {
  for (int i = 0; i < 8; i++) { h ^= (w >> (8 * i)) & 0xFFu; h *= 0x100000001B3ull; }
  return h;
}

uint64_t ora_server_checksum(const rgb_server_state *h) {
  uint64_t x = 0xCBF29CE484222325ull;
  x = fnv_word(x, h->current_term); x = fnv_word(x, h->commit_index);
  x = fnv_word(x, h->last_applied); x = fnv_word(x, h->last_index);
  x = fnv_word(x, h->last_term); x = fnv_word(x, h->last_written_index);
  x = fnv_word(x, h->last_written_term); x = fnv_word(x, h->snapshot_index);
  x = fnv_word(x, h->snapshot_term); x = fnv_word(x, h->first_index);
  uint64_t packed = (uint64_t)h->role | ((uint64_t)h->cond_reason << 8) | ((uint64_t)h->self << 16) |
                    ((uint64_t)h->n_members << 24) | ((uint64_t)h->voted_for << 32) |
                    ((uint64_t)h->leader_id << 40) | ((uint64_t)h->votes << 48) |
                    ((uint64_t)h->n_runs << 56);
  x = fnv_word(x, packed);
  uint64_t masks = (uint64_t)h->present_mask | ((uint64_t)h->voter_mask << 8) |
                   ((uint64_t)h->status_mask << 16) | ((uint64_t)h->self_nonvoter << 24) |
                   ((uint64_t)h->backoff_mask << 32);
  x = fnv_word(x, masks);
  x = fnv_word(x, h->pre_vote_token);
  x = fnv_word(x, h->pending_first);
  for (unsigned k = 0; k < h->n_pending_old && k < 2; k++) {
    x = fnv_word(x, h->pending_old[k][0]); x = fnv_word(x, h->pending_old[k][1]);
  }
  x = fnv_word(x, h->query_index);
  for (unsigned i = 0; i < h->n_members && i < RGB_MAX_MEMBERS; i++) x = fnv_word(x, h->peer_query_index[i]);
  x = fnv_word(x, (uint64_t)h->machine_version | ((uint64_t)h->effective_machine_version << 32));
  for (unsigned i = 0; i < h->n_members && i < RGB_MAX_MEMBERS; i++) {
    x = fnv_word(x, h->match_index[i]); x = fnv_word(x, h->next_index[i]);
    x = fnv_word(x, h->commit_index_sent[i]);
  }
  for (unsigned i = 0; i < h->n_runs && i < RGB_MAX_RUNS; i++) {
    x = fnv_word(x, h->run_start[i]); x = fnv_word(x, h->run_term[i]);
  }
  return x;
}

size_t ora_struct_size(int which) {
  switch (which) {
    case 0: return sizeof(rgb_msg);
    case 1: return sizeof(rgb_decision);
    case 2: return sizeof(rgb_rpc);
    case 3: return sizeof(rgb_server_state);
    case 4: return sizeof(rgb_leaderboard_row);
    case 5: return sizeof(rgb_config);
    default: return 0;
  }
}
