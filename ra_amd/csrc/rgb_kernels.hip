/*
 * rgb_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X): the batched Raft
 * append/vote transition and its support kernels.  Integer/index work, HBM-bound: no MFMA.
 *
 * rgb_tick_kernel<N>: one lane per inbound message; a tick carries at most one message per
 * server, so lanes never share state.  Per lane: 4x16-B loads of the message, 7x16-B loads
 * of the server's hot line, lazily the peers line (leader-side messages) and term-run probes
 * (log-matching repair), the transition itself in registers, then 16-B stores of whatever
 * changed and of the 64-B decision.  Outbound append_entries_rpc descriptors go to fixed slots
 * (message index x (N-1) + ordinal): no atomics, deterministic placement.
 *
 * Semantics restate (independently of oracle/) the reference clauses cited per function:
 *   src/ra_server.erl handle_leader/2 :530-1040, handle_candidate/2 :1043-1190,
 *   handle_pre_vote/2 :1192-1279, handle_follower/2 :1281-1657,
 *   handle_await_condition/2 :1916-1959, plus the ra_log cursor rules of src/ra_log.erl.
 */
#include <hip/hip_runtime.h>
#include "rgb_internal.h"

#define UNDEF 0xFFFFFFFFFFFFFFFFull
#define SLOT_NONE4 0xFu

/* Profiling knobs exist only in the -DRGB_PROFILE build (libra_gpu_batch_prof.so, used by tools/): the
 * product library has no run-time switch that changes what a tick computes.
 *   1 = no state write-back, 2 = no decision store, 8 = no hot-line load (zero state),
 *   16 = per-wave timestamps into dbg_buf, 32 = run-table probes answer from the last run (no dependent
 *   loads), 64 = the peers row is not loaded (zeros), 128 = the load generator emits benign traffic only
 *   (no term churn, no append_entries anomalies, no failed replies) */
#ifdef RGB_PROFILE
#define RGB_KNOB(dev, bit) (((dev).dbg & (bit)) != 0)
#else
#define RGB_KNOB(dev, bit) false
#endif

#ifndef RGB_TICK_BLOCK
#define RGB_TICK_BLOCK 64
#endif
#ifndef RGB_CLASS_MIN_WAVES
#define RGB_CLASS_MIN_WAVES(N) ((N) <= 5 ? 4 : 3)   /* the class-dispatch kernel: hot paths fit 128 VGPRs */
#endif
#ifndef RGB_MIN_WAVES
#define RGB_MIN_WAVES(N) 2   /* waves per SIMD the register allocator must leave room for */
#endif

namespace {

/* Messages per wavefront of a class.  The leader-side classes (append_entries_reply, append, pipeline_rpcs) take 32
 * when the peers row is one 128-byte line (3..5 members): their wavefronts fetch the peers rows WITH the hot rows --
 * one round trip, 8 lanes per line, into the LDS half the other 32 hot rows would have used -- instead of a second,
 * per-lane round trip once the clause code starts. */
#ifndef RGB_X_LEAD32
#define RGB_X_LEAD32 1
#endif
__host__ __device__ constexpr bool rgb_lead_class(int c) { return c == 1 || c == 3 || c == 4; }
/* classes whose kinds only use the first 32 bytes of their message records (include/ra_gpu_batch.h, "Field use by
 * kind": term, a, b at most; the two-range form of a written event re-reads its record): written, pipeline_rpcs,
 * request_vote, vote_result, await_timeout, snapshot_written, heartbeat_rpc, heartbeat_reply, consistent_query.
 * Their wavefronts request half of every record -- a third of a tick's messages, and a tick's time follows its bytes */
#ifndef RGB_X_HINT
#define RGB_X_HINT 1        /* the generator's steady-state bucketing hint (rgb_bucket_hinted); 0: A/B timing only */
#endif
#ifndef RGB_X_HALFMSG
#define RGB_X_HALFMSG 1
#endif
__host__ __device__ constexpr bool rgb_half_msg_class(int c) {
  return RGB_X_HALFMSG && (c == 2 || c == 4 || c == 5 || c == 6 || c == 7 || c == 11 || c == 12 || c == 13 || c == 14);
}
__host__ __device__ __forceinline__ constexpr u32 rgb_class_slice(int c, unsigned n_members) {
  return (RGB_X_LEAD32 && rgb_lead_class(c) && ((3u * n_members + 7u) & ~7u) == 16u) ? 32u : (u32)RGB_TICK_BLOCK;
}
/* TRAIN launches: the leader-side classes of groups of six to eight members (peers rows of 192 bytes) take 32 messages
 * too -- their wavefronts fetch the 32 peers rows cooperatively into LDS (12 lanes per row, 256 bytes of LDS each)
 * beside the 32 hot rows and the first line of the 32 run tables (16 KiB, what these kernels allocate anyway), where
 * every lane used to read its row from memory with eleven 16-byte loads of its own (round 5; BASELINE configs[4]) */
#ifndef RGB_X_NORPC
#define RGB_X_NORPC 0             /* 1: PROBE (breaks the output) -- no rpc record is stored: what the records' stores cost a tick */
#endif
#ifndef RGB_X_LEAD32_WIDE
#define RGB_X_LEAD32_WIDE 1
#endif
#ifndef RGB_TRAIN_RUNS_LDS
#define RGB_TRAIN_RUNS_LDS 1      /* the first line of a train wavefront's run tables comes into LDS with its rows */
#endif
__host__ __device__ __forceinline__ constexpr bool rgb_wide_peers(unsigned n_members) { return ((3u * n_members + 7u) & ~7u) == 24u; }
__host__ __device__ __forceinline__ constexpr u32 rgb_train_class_slice(int c, unsigned n_members) {
  /* (16 KiB of LDS per wavefront: the kernels built without the run-table line allocate less) */
  return (RGB_X_LEAD32 && RGB_X_LEAD32_WIDE && RGB_TRAIN_RUNS_LDS && rgb_lead_class(c) && rgb_wide_peers(n_members)) ? 32u
                                                                                                                  : rgb_class_slice(c, n_members);
}


__device__ __forceinline__ void store16_nt(void *p, ulonglong2 v) {
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  v4u d;
  d.x = (unsigned)v.x; d.y = (unsigned)(v.x >> 32); d.z = (unsigned)v.y; d.w = (unsigned)(v.y >> 32);
  /* the builtin, not inline assembly: the compiler must KNOW this is a 16-byte store to keep the next VALU write of
   * a data register the required wait states away from it (an asm statement gets no hazard protection: round 2
   * found decisions whose first dword was the NEXT store's index computation, in a few wavefronts per tick) */
  __builtin_nontemporal_store(d, reinterpret_cast<v4u *>(p));
}
/* state and rpc-record stores are plain (write-back L2): write-through and non-temporal flavours were
 * measured slower in round 1 */
#define ST16(ptr, val) do { *(ptr) = (val); } while (0)
#define ST8(ptr, val) do { *(ptr) = (val); } while (0)

/* One 56-byte rgb_rpc record (seven words, 8-byte aligned: records are 56 bytes apart) as three 16-byte stores and one
 * of 8 -- gfx950 takes a 16-byte store at any dword alignment -- instead of seven 8-byte stores (round 6: the records'
 * stores, 28 instructions per pipelining lane on the longest-lived wavefronts of a tick, cost 6-7 % of it by the
 * no-store probe; RGB_X_RPC_ST8 = 1 is the old form for A/B timing) */
#ifndef RGB_X_RPC_ST8
#define RGB_X_RPC_ST8 0
#endif
typedef unsigned long long rgb_u64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ void store_rpc(rgb_rpc *slot, u64 w0, u64 w1, u64 w2, u64 w3, u64 w4, u64 w5, u64 w6) {
  u64 *o = reinterpret_cast<u64 *>(slot);
#if RGB_X_RPC_ST8 || defined(RGB_HOST_EMULATION)
  ST8(o + 0, w0); ST8(o + 1, w1); ST8(o + 2, w2); ST8(o + 3, w3); ST8(o + 4, w4); ST8(o + 5, w5); ST8(o + 6, w6);
#else
  rgb_u64x2_a8 a, b, c;
  a.x = w0; a.y = w1; b.x = w2; b.y = w3; c.x = w4; c.y = w5;
  *reinterpret_cast<rgb_u64x2_a8 *>(o + 0) = a;
  *reinterpret_cast<rgb_u64x2_a8 *>(o + 2) = b;
  *reinterpret_cast<rgb_u64x2_a8 *>(o + 4) = c;
  ST8(o + 6, w6);
#endif
}
/* Deferred rpc records (round 6).  The records are OUTPUT: no later message reads them, yet stored where they are made
 * -- inside the clause code of the longest-lived wavefronts of a tick ({commands}: four records per lane) -- they stand
 * in front of the wavefront's publish twice: their issue, and their acknowledgements in the wait before it.  A train
 * wavefront keeps what it cannot recompute (32 bytes per record) in LDS and stores the records BEHIND its publish, like
 * its decisions.  Same records, same slots.  RGB_X_RPC_DEFER = 0: the direct form, A/B timing */
#ifndef RGB_X_RPC_DEFER
#define RGB_X_RPC_DEFER 1
#endif
template <class Lane>
__device__ __forceinline__ void emit_rpc(Lane &L, rgb_rpc *rpcs, u32 slot_base, unsigned ord, u32 msg_index, u64 w1,
                                         u64 rp_idx, u64 rp_term, u64 new_ni) {
  if (rpcs == nullptr || RGB_X_NORPC) return;
  if (RGB_X_RPC_DEFER && L.rpc_stash != nullptr && ord < 4u) {
    L.rpc_stash[2u * ord] = make_ulonglong2(w1, rp_idx);
    L.rpc_stash[2u * ord + 1u] = make_ulonglong2(rp_term, new_ni);
    return;
  }
  store_rpc(rpcs + (size_t)slot_base + ord, (u64)msg_index | ((u64)L.server << 32), w1, L.ct, rp_idx, rp_term, L.ci, new_ni);
}

/* -DRGB_X_MARK: comment markers around every class path in the assembly (tools/class_isa.py counts per class) */
#if defined(RGB_X_MARK) && !defined(RGB_HOST_EMULATION)
#define RGB_MARK(what, rank) asm volatile("; RGB_MARK " what " %0" ::"n"(rank));
#else
#define RGB_MARK(what, rank)
#endif

/* Asynchronous 16-byte global -> LDS copy (global_load_lds_dwordx4): lane l's bytes land at lds_base + 16 * l -- the
 * destination of the instruction is wave-uniform base + lane * 16 -- without passing through a register.  NT = data
 * that is read once.  glds_wait() = the issuing wave's copies have landed (a one-wave workgroup needs nothing else). */
enum { GLDS_DEFAULT = 0, GLDS_NT = 2, GLDS_SC1 = 16 };   /* cache policy: nt = read once; sc1 = served by L2 (L1 bypass) */
template <int POLICY>
__device__ __forceinline__ void glds16(const void *g, void *lds_base) {
#ifdef RGB_HOST_EMULATION
  memcpy(static_cast<char *>(lds_base) + 16 * emu::lane(), g, 16);
#else
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                   (__attribute__((address_space(3))) void *)lds_base, 16, 0, POLICY);
#endif
}
__device__ __forceinline__ void glds_wait() {
#ifndef RGB_HOST_EMULATION
  /* s_waitcnt vmcnt(0) (expcnt / lgkmcnt untouched) as a BUILTIN: the compiler's wait-count pass tracks LDS-DMA as a
   * pending vector-memory event and, not seeing a wait it understands, would put a vmcnt(0) in front of every later LDS
   * access -- on gfx950 that also waits for every global store issued meanwhile (the state write-back's
   * acknowledgements, ~2 us in front of the decision staging) */
  __builtin_amdgcn_s_waitcnt(0x0F70);
  asm volatile("" ::: "memory");
#endif
}

/* workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global
 * store (vmcnt(0)), which put the state write-back's acknowledgement on the decision store's path */
__device__ __forceinline__ void lds_barrier() {
#ifdef RGB_HOST_EMULATION
  __syncthreads();
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#endif
}

/* non-temporal 16-byte load: message records are read exactly once */
typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ ulonglong2 ld16(const ulonglong2 *p) {
  if (NT) {
    v2u64 v = __builtin_nontemporal_load(reinterpret_cast<const v2u64 *>(p));
    return make_ulonglong2(v.x, v.y);
  }
  return *p;
}

/* State loads of the clause code.  coh = the multi-tick train launch (rgb_train_kernel): another CU may have written
 * this server's rows earlier IN THE SAME LAUNCH, and a CU's vector L1 is never refreshed by another CU's stores, so the
 * load must be served by the XCD's L2: a relaxed agent-scope atomic load (global_load ... sc1, which bypasses L1 only;
 * the compiler tracks it like any load).  The per-tick kernels (coh = false) keep plain loads. */
__device__ __forceinline__ u64 ldg8(bool coh, const u64 *p) {
#ifndef RGB_HOST_EMULATION
  if (coh) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
  return *p;
}
__device__ __forceinline__ ulonglong2 ldg16(bool coh, const ulonglong2 *p) {
#ifndef RGB_HOST_EMULATION
  if (coh) {
    const u64 *q = reinterpret_cast<const u64 *>(p);
    const u64 x = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64 y = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_ulonglong2(x, y);
  }
#endif
  return *p;
}

__device__ __forceinline__ u64 pk_get(u64 pk, int sh, int w) { return (pk >> sh) & ((1ull << w) - 1ull); }
__device__ __forceinline__ u64 pk_set(u64 pk, int sh, int w, u64 v) {
  const u64 m = ((1ull << w) - 1ull) << sh;
  return (pk & ~m) | ((v << sh) & m);
}
__device__ __forceinline__ unsigned slot8to4(unsigned s) { return s >= 8u ? SLOT_NONE4 : s; }
__device__ __forceinline__ unsigned slot4to8(unsigned s) { return s >= 8u ? (unsigned)RGB_NONE : s; }

/* Everything one lane needs for one message: the server's hot line in registers, the message,
 * the effects being accumulated and the pending (uncommitted) log-table edits. */
template <bool COH, bool WC = false, bool SEQX = false>
struct LaneT {
  static constexpr bool coh = COH;   /* state loads must bypass the CU's L1 (train launch), see ldg8 */
  /* written events of more than two ranges (RGB_MF_SEQX) are served by the kind-generic kernel and by the written-only
   * kernel rgb_tick_kernel<N, RGB_MSG_WRITTEN, true> (256 registers each): the walk over a list of ranges keeps three
   * more message words alive through the handlers, which the per-tick class kernel -- at exactly its 128 registers --
   * answered with 216 spilled VGPRs.  rgb_submit routes the written class of a batch that holds such a record to the
   * written-only kernel (enqueue_rounds); any other specialised path that meets one reports RGB_F_UNHANDLED. */
  static constexpr bool seqx_ok = SEQX;
  /* WC (groups of six and more members: the kernels that have the registers): the run a table walk ended in is
   * remembered -- number, start, term, start of the next run -- and the next look-up of the same message tries it
   * first.  One append_entries_rpc asks has_log_entry_or_snapshot(prev), drop_existing (run of the first entry, its
   * term, its end), fetch_term(last_applied): on the log-matching repair workload (configs[4]: 1 024-entry backlogs
   * over 3-6 term boundaries, every prev_log_index inside the backlog) each of them walked the table again from its
   * newest run, 7.6 dependent round trips per lane.  The table rows a walk reads are never written before the commit,
   * so what is remembered stays true for the whole message. */
  static constexpr bool walk_cache = WC;
  int wc_k;
  u64 wc_start, wc_term, wc_next;
  /* hot line */
  u64 ct, ci, la, li, lt, lwi, lwt, pk, si, st, first, lrs, lrt, prs, prt, pend;
  /* cold words (qry row), loaded only by the election kinds */
  u64 token, macver;
  /* message */
  u32 server, n_entries, n_run0;
  unsigned kind, from, mflags, gap;
  u64 term, a, b, c, run0_term, run1_term;
  /* device */
  const u64 *seqx;       /* RGB_MF_SEQX: the launch's range list (first, last pairs), or null */
  u32 n_seqx;
  const u64 *runs;       /* this server's run table (start,term pairs) */
  u64 *peers;            /* this server's peers row                    */
  const ulonglong2 *peers_lds;   /* the same row in LDS (class kernel, leader-side classes, 128-byte rows), or null */
  unsigned peers_swz;            /* piece p of that row sits at position p ^ peers_swz */
  const ulonglong2 *runs_lds;    /* runs 0..RGB_RUNS_LDS-1 of the run table in LDS (train launches, leader-side classes:
                                  * fetched with the hot row; run k at position k ^ peers_swz), or null */
  ulonglong2 *rpc_stash;         /* train launches, leader-side classes: the message's first four rpc records wait HERE
                                  * -- 32 bytes each in the lane's own hot row in LDS, dead once the row is in registers --
                                  * until the wavefront has published (emit_rpc, rgb_tick_slice), or null */
  u32 max_runs;
  /* effects */
  u32 flags;
  u32 inv;
  bool has_reply;
  unsigned reply_to;
  u64 r_term, r_next, r_last, r_lterm;
  u64 w_first, w_last;
  bool vote_reqs;        /* {send_vote_requests,..}: the request fields ride in r_* */
  /* pending run-table edits: n_runs is the NEW count; the last `push_cnt` runs are not in
   * memory yet: (prs,prt) [when push_cnt == 2] then (lrs,lrt).  (prs,prt) always mirror run
   * n_runs-2 and (lrs,lrt) run n_runs-1: a lookup only touches memory for run n_runs-3 and older. */
  unsigned n_runs;
  unsigned push_cnt;
  bool cond_dirty;
  u64 cr0, cr1, cr2, cr3;
  /* sparse `pending` (packed-word bit PK_PENDX): the ranges below the newest one live in the qry row and are only
   * ever CUT by a message -- ra_seq:limit keeps what is below po_cut, ra_seq:floor / remove_prefix what is at or
   * above po_floor -- so the two bounds are collected here and applied at commit */
  u64 po_floor, po_cut;
  /* the server's peers row in registers (leader-side messages): match_index / next_index /
   * commit_index_sent per member, loaded with the hot line (one round trip), written back
   * word by word where the dirty masks say so.  Only indexes < N are ever touched (loops are
   * fully unrolled on the template parameter, so these arrays live in VGPRs). */
  u64 pmi[8], pni[8], pcs[8];
  unsigned dmi, dni, dcs;
  unsigned pdirty;   /* peers row in LDS (PL paths): bit w = word w of the row was written */
  unsigned dcs_ci;   /* peers whose commit_index_sent becomes L.ci at commit (pipelining): no per-peer copy is kept */
  bool peers_loaded;
  /* consistent-query heartbeats (cold row, loaded on demand) */
  u64 *qry_base;          /* uniform: the row address is recomputed where it is needed */
  u64 qself, qp[8];
  bool q_loaded;
  unsigned q_dirty;       /* bit 0: query_index, bit 1+i: peer slot i */
  unsigned hb_mask;
  unsigned cancel_mask;   /* RGB_F_CANCEL_SNAPSHOT_RETRY: backed-off peers contacted by make_all_rpcs */
#ifdef RGB_X_DECLINE_HIST
  const u64 *dbg_hist;    /* tools/train_decline_hist.py: counters of the hist build */
#endif
#ifdef RGB_PROFILE
  unsigned prof_nloads;   /* run-table words read by this lane */
  bool prof_noprobe;      /* knob 32: run-table probes answer from the last run (timing experiments only) */
#endif
  u64 hb_term, hb_qi, q_consensus;
};

template <int N, class Lane>
__device__ __forceinline__ void load_peers(Lane &L) {
  if (L.peers_loaded) return;
  constexpr int PS = (3 * N + 7) & ~7;
  u64 w[PS];
  if ((PS == 16 || PS == 24) && L.peers_lds != nullptr) {
    /* (192-byte rows -- six to eight members, train launches -- lie in 256-byte LDS rows, piece p at p ^ 2 swz) */
    const unsigned sw = PS == 24 ? L.peers_swz << 1 : L.peers_swz;
#pragma unroll
    for (int k = 0; k < PS / 2; ++k) {
      if (2 * k < 3 * N) { ulonglong2 v = L.peers_lds[(unsigned)k ^ sw]; w[2 * k] = v.x; w[2 * k + 1] = v.y; }
    }
  } else {
    const ulonglong2 *pp = reinterpret_cast<const ulonglong2 *>(L.peers);
#pragma unroll
    for (int k = 0; k < PS / 2; ++k) {
      if (2 * k < 3 * N) { ulonglong2 v = ldg16(L.coh, pp + k); w[2 * k] = v.x; w[2 * k + 1] = v.y; }
    }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) { L.pmi[i] = w[PEER_MI(i, N)]; L.pni[i] = w[PEER_NI(i, N)]; L.pcs[i] = w[PEER_CS(i, N)]; }
  L.peers_loaded = true;
}

template <int N>
__device__ __forceinline__ u64 peer_get(const u64 (&a)[8], unsigned p) {
  u64 r = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) r = ((unsigned)i == p) ? a[i] : r;
  return r;
}
template <int N>
__device__ __forceinline__ void peer_set(u64 (&a)[8], unsigned &dirty, unsigned p, u64 v) {
#pragma unroll
  for (int i = 0; i < N; ++i)
    if ((unsigned)i == p) { a[i] = v; dirty |= 1u << i; }
}

/* The peers row of a leader-side message handled by the class kernel lives in LDS (PL = true; fetched with the hot
 * row, piece p at position p ^ peers_swz): its words are read and written THERE, the register arrays pmi/pni/pcs
 * (30 VGPRs for five members) do not exist on those paths.  PL = false: the arrays, loaded from memory. */
template <class Lane>
__device__ __forceinline__ unsigned prow_at(const Lane &L, unsigned w) { return (((w >> 1) ^ L.peers_swz) << 1) | (w & 1u); }
template <class Lane>
__device__ __forceinline__ u64 prow_get(const Lane &L, unsigned w) {
  return reinterpret_cast<const u64 *>(L.peers_lds)[prow_at(L, w)];
}
template <class Lane>
__device__ __forceinline__ void prow_set(Lane &L, unsigned w, u64 v) {
  const_cast<u64 *>(reinterpret_cast<const u64 *>(L.peers_lds))[prow_at(L, w)] = v;
  L.pdirty |= 1u << w;
}
template <int N, bool PL, class Lane> __device__ __forceinline__ u64 mi_get(const Lane &L, int i) { return PL ? prow_get(L, PEER_MI(i, N)) : L.pmi[i]; }
template <int N, bool PL, class Lane> __device__ __forceinline__ u64 ni_get(const Lane &L, int i) { return PL ? prow_get(L, PEER_NI(i, N)) : L.pni[i]; }
template <int N, bool PL, class Lane> __device__ __forceinline__ u64 cs_get(const Lane &L, int i) { return PL ? prow_get(L, PEER_CS(i, N)) : L.pcs[i]; }
template <int N, bool PL, class Lane> __device__ __forceinline__ void ni_set(Lane &L, int i, u64 v) {
  if (PL) prow_set(L, PEER_NI(i, N), v); else { L.pni[i] = v; L.dni |= 1u << i; }
}
/* run-time peer index (the sender of a reply) */
template <int N, bool PL, class Lane> __device__ __forceinline__ u64 mi_of(const Lane &L, unsigned p) { return PL ? (p < (unsigned)N ? prow_get(L, PEER_MI(p, N)) : 0) : peer_get<N>(L.pmi, p); }
template <int N, bool PL, class Lane> __device__ __forceinline__ u64 ni_of(const Lane &L, unsigned p) { return PL ? (p < (unsigned)N ? prow_get(L, PEER_NI(p, N)) : 0) : peer_get<N>(L.pni, p); }
template <int N, bool PL, class Lane> __device__ __forceinline__ void mi_put(Lane &L, unsigned p, u64 v) {
  if (PL) { if (p < (unsigned)N) prow_set(L, PEER_MI(p, N), v); } else peer_set<N>(L.pmi, L.dmi, p, v);
}
template <int N, bool PL, class Lane> __device__ __forceinline__ void ni_put(Lane &L, unsigned p, u64 v) {
  if (PL) { if (p < (unsigned)N) prow_set(L, PEER_NI(p, N), v); } else peer_set<N>(L.pni, L.dni, p, v);
}

template <class Lane>
__device__ __forceinline__ bool range_nonempty(const Lane &L) { return L.first <= L.li; }

/* Runs of the table a leader-side train wavefront fetched into LDS with its hot rows (the first 128-byte line of the
 * row = the eight OLDEST runs: every in-memory run of a table of up to ten).  In a train every load of the table is an
 * L2 round trip (no L1, see ldg8) and the reply / {commands} wavefronts walk it in 99 % of the wavefronts (27.8 % of
 * their lanes, wave maximum 5 runs at the median, 8 at p90: profiles/r02_wave_timeline.txt) -- one round trip per
 * run, on the critical path of the wavefront's life.  The table is read-only for these kinds (a leader never truncates
 * its own log; a pushed run stays in registers until the commit). */
#define RGB_RUNS_LDS 8
#ifndef RGB_TRAIN_RUNS_LDS     /* (default set in front of rgb_train_class_slice) */
#define RGB_TRAIN_RUNS_LDS 1
#endif
/* (start, term) of in-memory run k */
template <class Lane>
__device__ __forceinline__ ulonglong2 run_pair(const Lane &L, int k) {
  if (L.runs_lds != nullptr && k < RGB_RUNS_LDS) return L.runs_lds[(unsigned)k ^ L.peers_swz];
  return ldg16(L.coh, reinterpret_cast<const ulonglong2 *>(L.runs) + k);
}
/* word i of this server's in-memory run table: (start, term) of run k at words 2k, 2k+1 */
template <class Lane>
__device__ __forceinline__ u64 run_word(const Lane &L, int i) {
#ifdef RGB_PROFILE
  const_cast<Lane &>(L).prof_nloads += 1;
#endif
  if (L.runs_lds != nullptr && i < 2 * RGB_RUNS_LDS) {
    const ulonglong2 v = L.runs_lds[(unsigned)(i >> 1) ^ L.peers_swz];
    return (i & 1) ? v.y : v.x;
  }
  return ldg8(L.coh, L.runs + i);
}

/* The in-memory runs (n_runs-3 and older) searched newest first for the one that holds idx: its number and term, -1 if
 * idx is below the oldest.  A run's (start, term) is one 16-byte load and the next older run is requested before the
 * current one is examined, so the walk overlaps its round trips two deep with two live pairs (requesting four at a
 * time measured 15 % slower per tick: the extra live registers spilled in the clause code around every call site). */
template <class Lane>
__device__ __forceinline__ int run_search(const Lane &L, u64 idx, u64 &term) {
  int k = (int)L.n_runs - 3;
  if (k < 0) return -1;
  if (Lane::walk_cache && L.wc_k >= 0 && L.wc_k <= k && idx >= L.wc_start && idx < L.wc_next) {
    term = L.wc_term;                                   /* the run the previous walk of this message ended in */
    return L.wc_k;
  }
  ulonglong2 cur = run_pair(L, k);
  u64 upper = L.prs;                                    /* start of run k + 1 (run n_runs - 2 is mirrored in the row) */
#pragma unroll 1
  for (; k >= 0; --k) {
    const ulonglong2 nxt = run_pair(L, k > 0 ? k - 1 : 0);
#ifdef RGB_PROFILE
    const_cast<Lane &>(L).prof_nloads += 2u;
#endif
    if (idx >= cur.x) {
      term = cur.y;
      if (Lane::walk_cache) {
        Lane &M = const_cast<Lane &>(L);
        M.wc_k = k; M.wc_start = cur.x; M.wc_term = cur.y; M.wc_next = upper;
      }
      return k;
    }
    upper = cur.x;
    cur = nxt;
  }
  return -1;
}

/* ra_log:fetch_term/2 (src/ra_log.erl:1186-1200): defined only inside the range */
template <class Lane>
__device__ __forceinline__ u64 fetch_term(const Lane &L, u64 idx) {
  if (!(range_nonempty(L) && idx >= L.first && idx <= L.li)) return UNDEF;
  if (idx >= L.lrs) return L.lrt;
#ifdef RGB_PROFILE
  if (L.prof_noprobe) return L.lrt;
#endif
  if (L.n_runs >= 2 && idx >= L.prs) return L.prt;
  /* runs n_runs-3 and older are all in memory (at most the newest two are pending) */
  u64 term = UNDEF;
  run_search(L, idx, term);
  return term;
}

/* ra_server:fetch_term/2 with the snapshot fallback (src/ra_server.erl:3185-3196) */
template <class Lane>
__device__ __forceinline__ u64 srv_fetch_term(const Lane &L, u64 idx) {
  u64 t = fetch_term(L, idx);
  if (t == UNDEF && L.si != UNDEF && L.si == idx) return L.st;
  return t;
}

/* index of the run holding idx (largest k with start_k <= idx), -1 if none.  Only called
 * before any edit of this message is pending. */
template <class Lane>
__device__ __forceinline__ int find_run(const Lane &L, u64 idx) {
  if (L.n_runs == 0) return -1;
  if (idx >= L.lrs) return (int)L.n_runs - 1;
#ifdef RGB_PROFILE
  if (L.prof_noprobe) return (int)L.n_runs - 1;
#endif
  if (L.n_runs >= 2 && idx >= L.prs) return (int)L.n_runs - 2;
  u64 term;
  return run_search(L, idx, term);
}

/* ra_log:last_index_term/1 (src/ra_log.erl:830-835) is simply (L.li, L.lt): an empty range
 * keeps them equal to the snapshot's, see rgb_server_state. */

/* ra_log:next_index/1 (src/ra_log.erl:1166-1174) */
template <class Lane>
__device__ __forceinline__ u64 next_log_index(const Lane &L) {
  if (range_nonempty(L)) return L.li + 1;
  if (L.si != UNDEF) return L.si + 1;
  return 0;
}

enum { HLE_OK = 0, HLE_MISMATCH = 1, HLE_MISSING = 2 };
/* has_log_entry_or_snapshot/3 (src/ra_server.erl:3168-3183) */
template <class Lane>
__device__ __forceinline__ int has_log_entry_or_snapshot(const Lane &L, u64 idx, u64 term) {
  u64 t = fetch_term(L, idx);
  if (t == UNDEF) {
    if (L.si != UNDEF && L.si == idx) return L.st == term ? HLE_OK : HLE_MISMATCH;
    return HLE_MISSING;
  }
  return t == term ? HLE_OK : HLE_MISMATCH;
}

/* ---- state word helpers ---- */
template <class Lane>
__device__ __forceinline__ unsigned role_of(const Lane &L) { return (unsigned)pk_get(L.pk, PK_ROLE_SH, 3); }
template <class Lane>
__device__ __forceinline__ unsigned self_of(const Lane &L) { return (unsigned)pk_get(L.pk, PK_SELF_SH, 4); }
template <class Lane>
__device__ __forceinline__ bool present(const Lane &L, unsigned i) {
  return i < 8u && ((L.pk >> (PK_PRESENT_SH + i)) & 1ull);
}
template <class Lane>
__device__ __forceinline__ bool voter(const Lane &L, unsigned i) { return (L.pk >> (PK_VOTER_SH + i)) & 1ull; }
template <class Lane>
__device__ __forceinline__ bool status_normal(const Lane &L, unsigned i) { return (L.pk >> (PK_STATUS_SH + i)) & 1ull; }

/* role change; become(follower,..) resets every peer status to normal
 * (src/ra_server.erl:2183-2192) */
template <class Lane>
__device__ __forceinline__ void set_role(Lane &L, unsigned role) {
  unsigned old = role_of(L);
  if (old != role) L.flags |= RGB_F_ROLE_CHANGED;
  if (role == RGB_ROLE_FOLLOWER && old != RGB_ROLE_FOLLOWER) {
    L.pk = pk_set(L.pk, PK_STATUS_SH, 8, 0xFF);
    L.pk = pk_set(L.pk, PK_BACKOFF_SH, 1, 0);         /* status => normal for every peer */
  }
  if (role != RGB_ROLE_AWAIT_CONDITION) L.pk &= ~((3ull << PK_COND_SH) | (1ull << PK_CONDTO_SH));   /* RGB_COND_NONE */
  L.pk = pk_set(L.pk, PK_ROLE_SH, 3, role);
}

/* update_term_and_voted_for/3 (src/ra_server.erl:3041-3058); voted4 is a 4-bit slot */
template <class Lane>
__device__ __forceinline__ void update_term_and_voted_for(Lane &L, u64 term, unsigned voted4) {
  unsigned cur = (unsigned)pk_get(L.pk, PK_VOTED_SH, 4);
  if (term == L.ct && voted4 == cur) return;
  L.flags |= RGB_F_PERSIST;
  L.ct = term;
  L.pk = pk_set(L.pk, PK_VOTED_SH, 4, voted4);
  /* reset_query_index/1 :3769-3773: only when some peer query_index is non-zero */
  if (pk_get(L.pk, PK_QPEER_SH, 1)) {
    /* the cleared bit IS the reset: readers take the peers as zero, the commit stage zeroes the
     * row when the bit went from set to clear */
    L.pk = pk_set(L.pk, PK_QPEER_SH, 1, 0);
    if (L.q_loaded) {      /* never on the hot kinds: their paths do not load the row */
#pragma unroll
      for (int i = 0; i < 8; ++i) L.qp[i] = 0;
    }
  }
}

template <class Lane>
__device__ __forceinline__ u64 *qry_row(const Lane &L) { return L.qry_base + (size_t)L.server * RGB_QRY_WORDS; }

/* the qry row: query_index | per-peer query_index */
template <int N, class Lane>
__device__ __forceinline__ void load_qry(Lane &L) {
  if (L.q_loaded) return;
  const ulonglong2 *qp = reinterpret_cast<const ulonglong2 *>(qry_row(L));
  u64 w[N + 2];
#pragma unroll
  for (int k = 0; k < (N + 2) / 2; ++k) { ulonglong2 v = ldg16(L.coh, qp + k); w[2 * k] = v.x; w[2 * k + 1] = v.y; }
  L.qself = w[0];
  if (pk_get(L.pk, PK_QPEER_SH, 1)) {
#pragma unroll
    for (int i = 0; i < N; ++i) L.qp[i] = w[1 + i];
  }
  L.q_loaded = true;
}

/* heartbeat_reply/2 :3727-3729 cast to the sender of the #heartbeat_rpc{} */
template <class Lane>
__device__ __forceinline__ void heartbeat_reply(Lane &L, u64 term, u64 query_index, unsigned to8) {
  L.has_reply = true;
  L.flags |= RGB_F_REPLY | RGB_F_REPLY_HEARTBEAT;
  L.r_term = term; L.r_next = query_index; L.r_last = 0; L.r_lterm = 0;
  L.reply_to = to8;
}

template <int N, class Lane>
__device__ __forceinline__ unsigned n_other_members(const Lane &L) {
  const unsigned present = (unsigned)pk_get(L.pk, PK_PRESENT_SH, 8) & ((1u << N) - 1u);
  return (unsigned)__popc(present & ~(1u << (unsigned)pk_get(L.pk, PK_SELF_SH, 4)));
}

/* heartbeat_rpc_effects/4 + heartbeat_rpc_effect_for_peer/5 :3775-3795 */
template <int N, class Lane>
__device__ __forceinline__ void heartbeat_rpc_effects(Lane &L, u64 query_index) {
  const unsigned self = (unsigned)pk_get(L.pk, PK_SELF_SH, 4);
  const unsigned present = (unsigned)pk_get(L.pk, PK_PRESENT_SH, 8);
  const unsigned status = (unsigned)pk_get(L.pk, PK_STATUS_SH, 8);
  unsigned mask = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if ((unsigned)i == self || !((present >> i) & 1u) || !((status >> i) & 1u)) continue;
    if (L.qp[i] < query_index) mask |= 1u << i;
  }
  if (mask) {
    L.flags |= RGB_F_SEND_HEARTBEATS;
    L.hb_mask |= mask;
    L.hb_term = L.ct; L.hb_qi = query_index;
  }
}

/* peers (of `candidates`) whose query_index is below the row's own query_index, straight from
 * memory: the election paths that call it must not carry the row in registers */
__device__ __forceinline__ unsigned heartbeat_targets(const u64 *qry, unsigned candidates, bool peers_zero, u64 *qself,
                                                      bool coh) {
  const u64 q0 = ldg8(coh, qry);
  unsigned mask = 0;
  for (unsigned i = 0; i < 8; ++i) {
    if (!((candidates >> i) & 1u)) continue;
    const u64 qi = peers_zero ? 0 : ldg8(coh, qry + 1 + i);
    if (qi < q0) mask |= 1u << i;
  }
  *qself = q0;
  return mask;
}

/* update_heartbeat_rpc_effects/1 :3731-3747 (the waiting queue lives on the host) */
template <int N, class Lane>
__device__ __forceinline__ void update_heartbeat_rpc_effects(Lane &L) {
  if (n_other_members<N>(L) == 0) { L.flags |= RGB_F_QUERY_APPLY; return; }
  if (!pk_get(L.pk, PK_QSELF_SH, 1)) return;      /* query_index == 0: no peer can be below it */
  const unsigned self = (unsigned)pk_get(L.pk, PK_SELF_SH, 4);
  const unsigned cand = (unsigned)pk_get(L.pk, PK_PRESENT_SH, 8) & (unsigned)pk_get(L.pk, PK_STATUS_SH, 8) &
                        ((1u << N) - 1u) & ~(1u << self);
  u64 q0;
  const unsigned mask = heartbeat_targets(qry_row(L), cand, !pk_get(L.pk, PK_QPEER_SH, 1), &q0, L.coh);
  if (mask) {
    L.flags |= RGB_F_SEND_HEARTBEATS;
    L.hb_mask |= mask;
    L.hb_term = L.ct; L.hb_qi = q0;
  }
}

/* make_heartbeat_rpc_effects/2 :3749-3767 */
template <int N, class Lane>
__device__ __forceinline__ void make_heartbeat_rpc_effects(Lane &L) {
  if (n_other_members<N>(L) == 0) { L.flags |= RGB_F_QUERY_APPLY; return; }
  load_qry<N>(L);
  L.qself += 1; L.q_dirty |= 1u;
  L.pk = pk_set(L.pk, PK_QSELF_SH, 1, 1);
  heartbeat_rpc_effects<N>(L, L.qself);
  L.hb_term = L.ct; L.hb_qi = L.qself;            /* the host queues the query under this index */
}

template <int N> __device__ __forceinline__ u64 agreed_commit(const u64 (&v)[N], const bool (&use)[N], int n);

/* heartbeat_rpc_quorum/3 :3797-3814, update_peer_query_index/3 :3816-3829,
 * get_current_query_quorum/1 :3831-3832 over query_indexes/1 :3659-3669 */
template <int N, class Lane>
__device__ __forceinline__ void heartbeat_rpc_quorum(Lane &L, u64 new_qi, unsigned peer) {
  const unsigned self = (unsigned)pk_get(L.pk, PK_SELF_SH, 4);
  const unsigned present = (unsigned)pk_get(L.pk, PK_PRESENT_SH, 8);
  const unsigned voters = (unsigned)pk_get(L.pk, PK_VOTER_SH, 8);
  load_qry<N>(L);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if ((unsigned)i == peer && ((present >> i) & 1u) && new_qi > L.qp[i]) {
      L.qp[i] = new_qi; L.q_dirty |= 2u << i;
      L.pk = pk_set(L.pk, PK_QPEER_SH, 1, 1);
    }
  }
  u64 v[N]; bool use[N]; int n = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if ((unsigned)i == self) { v[i] = L.qself; use[i] = true; n += 1; }
    else {
      use[i] = ((present >> i) & 1u) && ((voters >> i) & 1u);
      v[i] = L.qp[i];
      n += use[i] ? 1 : 0;
    }
  }
  L.flags |= RGB_F_QUERY_QUORUM;
  L.q_consensus = agreed_commit<N>(v, use, n);
}
/* update_term/2 (src/ra_server.erl:3060-3064) */
template <class Lane>
__device__ __forceinline__ void update_term(Lane &L, u64 term) {
  if (term != UNDEF && term > L.ct) update_term_and_voted_for(L, term, SLOT_NONE4);
}
template <class Lane>
__device__ __forceinline__ void set_leader_id(Lane &L, unsigned l4) {
  if ((unsigned)pk_get(L.pk, PK_LEADER_SH, 4) != l4) L.flags |= RGB_F_LEADER_CHANGED;
  L.pk = pk_set(L.pk, PK_LEADER_SH, 4, l4);
}

/* append_entries_reply/3 (src/ra_server.erl:3624-3631) */
template <class Lane>
__device__ __forceinline__ void aer_reply(Lane &L, u64 term, bool success, unsigned to8) {
  L.has_reply = true;
  L.flags |= RGB_F_REPLY | (success ? RGB_F_REPLY_SUCCESS : 0u);
  L.r_term = term; L.r_next = L.li + 1; L.r_last = L.lwi; L.r_lterm = L.lwt;
  L.reply_to = to8;
}
template <class Lane>
__device__ __forceinline__ void vote_reply(Lane &L, u64 term, bool granted, unsigned to8) {
  L.has_reply = true;
  L.flags |= RGB_F_REPLY | RGB_F_REPLY_VOTE | (granted ? RGB_F_REPLY_SUCCESS : 0u);
  L.r_term = term; L.r_next = 0; L.r_last = 0; L.r_lterm = 0;
  L.reply_to = to8;
}

template <class Lane>
__device__ __forceinline__ void pre_vote_reply(Lane &L, u64 term, u64 token, bool granted, unsigned to8) {
  L.has_reply = true;
  L.flags |= RGB_F_REPLY | RGB_F_REPLY_PRE_VOTE | (granted ? RGB_F_REPLY_SUCCESS : 0u);
  L.r_term = term; L.r_next = token; L.r_last = 0; L.r_lterm = 0;
  L.reply_to = to8;
}

/* required_quorum/1 (src/ra_server.erl:3996-3999), count_voters/1 :4001-4009 */
template <class Lane>
__device__ __forceinline__ unsigned required_quorum(const Lane &L) {
  unsigned voters = __popc((unsigned)(pk_get(L.pk, PK_PRESENT_SH, 8) & pk_get(L.pk, PK_VOTER_SH, 8)));
  return voters / 2 + 1;
}

/* apply_to/5 (src/ra_server.erl:3250-3282): only the cursor moves on the device */
template <class Lane>
__device__ __forceinline__ bool apply_to(Lane &L, u64 upto) {
  if (upto > L.la) {
    u64 to = L.li < upto ? L.li : upto;
    if (to >= L.la + 1) { L.la = to; return true; }
  }
  return false;
}

/* evaluate_commit_index_follower/2 (src/ra_server.erl:2246-2280) */
template <class Lane>
__device__ __forceinline__ void evaluate_commit_index_follower(Lane &L) {
  if (pk_get(L.pk, PK_LEADER_SH, 4) == SLOT_NONE4) return;
  u64 at = L.li < L.ci ? L.li : L.ci;
  if (apply_to(L, at)) L.flags |= RGB_F_APPLIED | RGB_F_AUX_EVAL;
}

/* ---- log edits (register side; memory is touched at commit) ---- */

/* append one segment [s..e] of term t after the current last run */
template <class Lane>
__device__ __forceinline__ void push_segment(Lane &L, u64 s, u64 t) {
  if (L.n_runs > 0 && L.lrt == t) return;            /* extends the last run */
  L.prs = L.lrs; L.prt = L.lrt;                      /* the old last run is now run n-2 */
  L.push_cnt += 1;
  L.n_runs += 1;
  L.lrs = s; L.lrt = t;
}

/* cut the table so that its last run is the one holding `keep_idx` (largest start <=
 * keep_idx); runs starting above it disappear.  keep_term = term_at(keep_idx) if known. */
template <class Lane>
__device__ __forceinline__ void truncate_runs_to(Lane &L, u64 keep_idx) {
  int k = find_run(L, keep_idx);
  if (k < 0) { L.n_runs = 0; return; }
  if ((unsigned)k != L.n_runs - 1) {
    if ((unsigned)k == L.n_runs - 2) { L.lrs = L.prs; L.lrt = L.prt; }
    else { L.lrs = run_word(L, 2 * k); L.lrt = run_word(L, 2 * k + 1); }
    L.n_runs = (unsigned)k + 1;
    if (k >= 1) { L.prs = run_word(L, 2 * k - 2); L.prt = run_word(L, 2 * k - 1); }   /* the new run n-2 */
  }
}

/* ra_log `pending` (src/ra_log.erl:126): on this path always the contiguous tail
 * [pend .. last_index] of indexes handed to the WAL and not yet confirmed; empty is held as
 * last_index + 1. */
template <class Lane>
__device__ __forceinline__ bool pend_nonempty(const Lane &L) { return range_nonempty(L) && L.pend <= L.li; }
template <class Lane>
__device__ __forceinline__ void pend_canon(Lane &L) { if (!pend_nonempty(L)) L.pend = L.li + 1; }
/* ra_seq:limit(CeilExcl - 1, Pend) / ra_seq:floor(Floor, Pend) on the ranges below the newest one */
template <class Lane>
__device__ __forceinline__ void pend_old_limit(Lane &L, u64 ceil_excl) { if (ceil_excl < L.po_cut) L.po_cut = ceil_excl; }
template <class Lane>
__device__ __forceinline__ void pend_old_floor(Lane &L, u64 floor_incl) { if (floor_incl > L.po_floor) L.po_floor = floor_incl; }

/* ra_log:write/2 (src/ra_log.erl:547-599, range update :1618-1623) of entries k0..n-1.
 * Returns an RGB_INV_* code, 0 on success.  Validates before editing. */
template <class Lane>
__device__ __forceinline__ int log_write(Lane &L, u32 k0) {
  const u64 base = L.a + 1 + (u64)L.gap;
  const u64 fst = base + k0;
  const u64 lst = base + (L.n_entries - 1);
  const bool had_range = range_nonempty(L);
  if (had_range && !(fst <= L.li + 1)) return RGB_INV_WRITE_INTEGRITY;
  if (fst == 0) return RGB_INV_WRITE_INTEGRITY;
  /* NewRange = ra_range:new(Start, LastIdx) (src/ra_log.erl:1617-1622), and new/2 with Start > End is `undefined`
   * (src/ra_range.erl:41-50): a write that ends below the start of a sparse range leaves NO range; last_index_term/1
   * (:831-835) is the snapshot's from then on */
  const bool range_lost = had_range && L.first > lst;
  if (range_lost && L.si == UNDEF) return RGB_INV_WRITE_INTEGRITY;
  u64 lwi = fst - 1 < L.lwi ? fst - 1 : L.lwi;
  u64 lwt;
  if (lwi == L.lwi) lwt = L.lwt;
  else if (L.si != UNDEF && L.si == lwi) lwt = L.st;
  else if (lwi == 0) lwt = 0;
  else {
    lwt = fetch_term(L, lwi);
    if (lwt == UNDEF) return RGB_INV_LAST_WRITTEN_TERM;
  }
  if (!had_range) {
    L.first = fst; L.n_runs = 0;
  } else if (fst <= L.li) {
    /* overwrite: runs starting at or after fst vanish */
    if (fst == 0 || fst <= L.first) L.n_runs = 0;
    else truncate_runs_to(L, fst - 1);
  }
  /* entries k0.. : first the rest of term run 0, then term run 1 */
  /* the range keeps its Start (src/ra_log.erl:1617-1622): indexes written below first_index stay
   * invisible, so the runs begin at max(fst, first_index) */
  const u64 lo = (had_range && L.first > fst) ? L.first : fst;
  if (range_lost) {
    /* the canonical undefined range behind a snapshot: (last index, last term) = the snapshot's, no runs */
    L.n_runs = 0; L.push_cnt = 0;
    L.li = L.si; L.lt = L.st; L.first = L.si + 1;
  } else {
    if (k0 < L.n_run0) {
      const u64 s1 = base + L.n_run0;
      if (L.n_run0 == L.n_entries || lo < s1) push_segment(L, lo, L.run0_term);
      if (L.n_run0 < L.n_entries) push_segment(L, s1 > lo ? s1 : lo, L.run1_term);
    } else {
      push_segment(L, lo, L.run1_term);
    }
    L.li = lst;
    L.lt = (L.n_entries - 1) < L.n_run0 ? L.run0_term : L.run1_term;
  }
  L.lwi = lwi; L.lwt = lwt;
  if (fst < L.pend) L.pend = fst;      /* ra_seq:limit(FstIdx-1, Pend0) :583 + ra_seq:append per entry :1610 */
  pend_old_limit(L, fst);
  return 0;
}

/* ra_log:set_last_index/2 (src/ra_log.erl:842-893) */
template <class Lane>
__device__ __forceinline__ int log_set_last_index(Lane &L, u64 idx) {
  u64 t = fetch_term(L, idx);
  bool snap_is_idx = (L.si != UNDEF && L.si == idx);
  if (t == UNDEF && !snap_is_idx) return RGB_INV_SET_LAST_INDEX_NOT_FOUND;
  if (snap_is_idx) {
    /* ra_range:limit(Idx+1, Range) (src/ra_range.erl:80-91) */
    if (range_nonempty(L)) {
      if (idx + 1 <= L.first) { L.n_runs = 0; L.li = idx; L.first = idx + 1; }
      else if (idx + 1 <= L.li) { truncate_runs_to(L, idx); L.li = idx; }
    }
    if (!range_nonempty(L)) L.li = L.si;
    L.lt = L.st;
    L.lwi = L.si; L.lwt = L.st;
    if (idx + 1 < L.pend) L.pend = idx + 1;   /* pending = ra_seq:limit(Idx, Pend0) :868 */
    pend_old_limit(L, idx + 1);
    pend_canon(L);
    return 0;
  }
  u64 lwi = idx < L.lwi ? idx : L.lwi;
  u64 lwt;
  if (L.si != UNDEF && L.si == lwi) lwt = L.st;
  else lwt = fetch_term(L, lwi);
  if (lwt == UNDEF) return RGB_INV_LAST_WRITTEN_TERM;
  if (idx + 1 <= L.li) { truncate_runs_to(L, idx); L.li = idx; }
  L.lt = t;
  L.lwi = lwi; L.lwt = lwt;
  if (idx + 1 < L.pend) L.pend = idx + 1;     /* pending = ra_seq:limit(Idx, Pend0) :891 */
  pend_old_limit(L, idx + 1);
  pend_canon(L);
  return 0;
}

/* ra_log:handle_event({written,Term,WrittenSeq}) (src/ra_log.erl:897-944) for a WrittenSeq of one range
 * [from..to] or two (RGB_MF_SEQ2: [w2s..w2e] below it).  The reference walks the sequence down one index at a time
 * (ra_seq:limit(Last-1), the retry of :931-943) until either the index's term is Term (first clause: last_written
 * moves there) or the index is undefined and at/below the snapshot (second clause: only `pending` is trimmed).
 * Run-wise here: c1 = the highest index of the sequence inside the log range whose run has term Term, c2 = the
 * highest index of the sequence outside the range and at/below the snapshot; the higher one decides and W_eff =
 * the sequence limited to it.  `pending` loses the written prefix (ra_seq:remove_prefix/2, drop_prefix :278-291):
 * every pending index at or below the stop index must be IN W_eff, else {error, not_prefix} -- a resend request
 * in the first clause (:917-919, cursors unchanged), a failed match in the second (:929).
 * Returns an RGB_INV_* code; `changed` = last_written moved. */

/* c1 over one range [lo..hi] of the sequence */
template <class Lane>
__device__ __forceinline__ bool written_c1(const Lane &L, u64 term, u64 from, u64 to, u64 &c1) {
  const u64 hi = to < L.li ? to : L.li;
  const u64 lo = from > L.first ? from : L.first;
  if (hi < lo) return false;
  /* walk runs from the newest: run k covers [start_k, end_k] */
  u64 end = L.li;
  for (int k = (int)L.n_runs - 1; k >= 0; --k) {
    u64 s, t;
    if ((unsigned)k == L.n_runs - 1) { s = L.lrs; t = L.lrt; }
    else if ((unsigned)k == L.n_runs - 2) { s = L.prs; t = L.prt; }
    else {
#ifdef RGB_PROFILE
      if (L.prof_noprobe) break;
#endif
      /* L2-served in a train launch: the table's older runs were pushed there by an earlier tick of the same launch */
      const ulonglong2 r = ldg16(L.coh, reinterpret_cast<const ulonglong2 *>(L.runs) + k);
#ifdef RGB_PROFILE
      const_cast<Lane &>(L).prof_nloads += 2;
#endif
      s = r.x; t = r.y;
    }
    const u64 rs = s < L.first ? L.first : s;
    if (rs <= hi && end >= lo && t == term) {
      const u64 idx = end < hi ? end : hi;
      if (idx >= lo && idx >= rs) { c1 = idx; return true; }
    }
    if (s <= lo) break;
    end = s - 1;
  }
  return false;
}
/* c2 over one range of the sequence */
template <class Lane>
__device__ __forceinline__ bool written_c2(const Lane &L, bool in_range, u64 from, u64 to, u64 &c2) {
  if (L.si == UNDEF) return false;
  u64 u = to < L.si ? to : L.si;
  bool ok = u >= from;
  if (ok && in_range && u >= L.first && u <= L.li) {      /* inside the range: next one below it */
    ok = L.first > 0 && L.first - 1 >= from;
    u = L.first - 1;
  }
  if (ok) c2 = u;
  return ok;
}
/* range r of the written sequence, HIGHEST first: r = 0 the record's (a, b); r = 1 its second range (RGB_MF_SEQ2:
 * run0_term .. run1_term); r >= 2 the ranges of the launch's list (RGB_MF_SEQX: entries c .. c + n_entries - 1,
 * ascending and all below the record's two -- so the list is walked from its top) */
template <class Lane>
__device__ __forceinline__ void written_range(const Lane &L, u32 r, u64 from, u64 to, u64 &f, u64 &t) {
  if (r == 0) { f = from; t = to; }
  else if (r == 1) { f = L.run0_term; t = L.run1_term; }
  else {
    const u64 *e = L.seqx + 2u * ((u64)L.c + (u64)(L.n_entries - 1u - (r - 2u)));
    f = e[0]; t = e[1];
  }
}
/* the pending range [ps .. pe], up to the stop index c, lies inside ONE range of the written sequence (or is empty,
 * or entirely above c: it stays) */
template <class Lane>
__device__ __forceinline__ bool pend_range_written_many(const Lane &L, u64 ps, u64 pe, u64 c, u32 n_w, u64 from, u64 to) {
  if (ps > pe || ps > c) return true;
  const u64 e = pe < c ? pe : c;
#pragma unroll 1
  for (u32 r = 0; r < n_w; ++r) {
    u64 f, t;
    written_range(L, r, from, to, f, t);
    const u64 te = t < c ? t : c;
    if (f <= c && ps >= f && e <= te) return true;
  }
  return false;
}

/* ra_log:handle_event({written, Term, Seq}) (src/ra_log.erl:897-944) for a sequence of ANY number of ranges
 * (round 5: one, two inline, or more through the launch's range list) against a `pending` of up to three ranges:
 * the retry of :931-943 walks the sequence from its top until an index with the event's term (clause 1) or below
 * the snapshot (clause 2) is found; ra_seq:remove_prefix/2 (src/ra_seq.erl:144-147, 278-291) then asks every pending
 * range up to that index to lie inside one written range */
template <class Lane>
__device__ __forceinline__ int log_written_many(Lane &L, u64 term, u64 from, u64 to, bool &changed) {
  changed = false;
  const bool in_range = range_nonempty(L);
  const bool two = (L.mflags & (RGB_MF_SEQ2 | RGB_MF_SEQX)) != 0;
  u32 n_w = two ? 2u : 1u;
  if (L.mflags & RGB_MF_SEQX) {
    /* the list must be there and hold the entries the record names (a malformed record commits nothing) */
    if (L.seqx == nullptr || L.n_entries == 0u || (u64)L.c + (u64)L.n_entries > (u64)L.n_seqx) return RGB_INV_WRITTEN_SEQ_LIST;
    n_w += L.n_entries;
  }
  bool have1 = false; u64 c1 = 0;
  bool have2 = false; u64 c2 = 0;
  /* the highest range first, then the lower ones: one copy of the walk in the code */
#pragma unroll 1
  for (u32 r = 0; r < n_w; ++r) {
    u64 f, t;
    written_range(L, r, from, to, f, t);
    if (in_range && !have1) have1 = written_c1(L, term, f, t, c1);
    if (!have2) have2 = written_c2(L, in_range, f, t, c2);
  }
  const bool first_clause = have1 && (!have2 || c1 > c2);
  if (!first_clause && !have2) return 0;                  /* the sequence ran out: no change (:934-938) */
  const u64 c = first_clause ? c1 : c2;
  /* ra_seq:remove_prefix(W_eff, Pend) */
  bool prefix = true;
  const bool sparse = pk_get(L.pk, PK_PENDX_SH, 1) != 0;
  if (pend_nonempty(L)) prefix = pend_range_written_many(L, L.pend, L.li, c, n_w, from, to);
  if (sparse && prefix) {
    const u64 *q = qry_row(L);
    /* the old ranges as this message has cut them so far (nothing cuts them before a written event, but be exact) */
#pragma unroll 1
    for (int k = 0; k < 2 && prefix; ++k) {
      u64 ps = ldg8(L.coh, q + QRY_PEND_LO + 2 * k), pe = ldg8(L.coh, q + QRY_PEND_LO + 2 * k + 1);
      if (ps < L.po_floor) ps = L.po_floor;
      if (L.po_cut != UNDEF && pe >= L.po_cut) pe = L.po_cut - 1;
      if (L.po_cut == 0) continue;
      prefix = pend_range_written_many(L, ps, pe, c, n_w, from, to);
    }
  }
  if (!prefix) {
    if (first_clause) { L.flags |= RGB_F_RESEND_PENDING; return 0; }
    return RGB_INV_WRITTEN_NOT_PREFIX;
  }
  if (pend_nonempty(L)) {
    if (c + 1 > L.pend) L.pend = c + 1;
    pend_canon(L);
  }
  pend_old_floor(L, c + 1);
  if (first_clause) {
    changed = !(L.lwi == c && L.lwt == term);
    L.lwi = c; L.lwt = term;
  }
  return 0;
}

/* is the pending range [ps..pe], cut at the stop index c, covered by W_eff = ([w2s..w2e] u [from..to]) limited to c?
 * (a contiguous pending range cannot straddle the gap between the two written ranges) */
__device__ __forceinline__ bool pend_range_written(u64 ps, u64 pe, u64 c, bool two, u64 w2s, u64 w2e, u64 from, u64 to) {
  if (ps > pe || ps > c) return true;                     /* empty, or entirely above the stop index: stays */
  const u64 e = pe < c ? pe : c;
  const u64 te = to < c ? to : c, t2e = w2e < c ? w2e : c;
  if (from <= c && ps >= from && e <= te) return true;
  if (two && w2s <= c && ps >= w2s && e <= t2e) return true;
  return false;
}

template <class Lane>
__device__ __forceinline__ int log_written(Lane &L, u64 term, u64 from, u64 to, bool &changed) {
  /* more than two ranges (RGB_MF_SEQX, round 5): the general walk above -- its own branch, so that the one- and
   * two-range events every tick carries keep the registers they had (folded into this function the loop over a list
   * of ranges cost the per-tick class kernel 216 spilled VGPRs) */
  if (L.mflags & RGB_MF_SEQX) {
    if (Lane::seqx_ok) return log_written_many(L, term, from, to, changed);
    changed = false;
    L.flags |= RGB_F_UNHANDLED;                           /* (a specialised path: the record belongs in the generic kernel) */
    return 0;
  }
  changed = false;
  const bool in_range = range_nonempty(L);
  const bool two = (L.mflags & RGB_MF_SEQ2) != 0;
  const u64 w2s = L.run0_term, w2e = L.run1_term;         /* the lower written range rides in these fields */
  bool have1 = false; u64 c1 = 0;
  bool have2 = false; u64 c2 = 0;
  /* the upper range first, then (two-range sequences only) the lower one: one copy of the walk in the code */
#pragma unroll 1
  for (int r = 0; r < (two ? 2 : 1); ++r) {
    const u64 f = r ? w2s : from, t = r ? w2e : to;
    if (in_range && !have1) have1 = written_c1(L, term, f, t, c1);
    if (!have2) have2 = written_c2(L, in_range, f, t, c2);
  }
  const bool first_clause = have1 && (!have2 || c1 > c2);
  if (!first_clause && !have2) return 0;                  /* the sequence ran out: no change (:934-938) */
  const u64 c = first_clause ? c1 : c2;
  /* ra_seq:remove_prefix(W_eff, Pend) */
  bool prefix = true;
  const bool sparse = pk_get(L.pk, PK_PENDX_SH, 1) != 0;
  if (pend_nonempty(L)) prefix = pend_range_written(L.pend, L.li, c, two, w2s, w2e, from, to);
  if (sparse && prefix) {
    const u64 *q = qry_row(L);
    /* the old ranges as this message has cut them so far (nothing cuts them before a written event, but be exact) */
#pragma unroll 1
    for (int k = 0; k < 2 && prefix; ++k) {
      u64 ps = ldg8(L.coh, q + QRY_PEND_LO + 2 * k), pe = ldg8(L.coh, q + QRY_PEND_LO + 2 * k + 1);
      if (ps < L.po_floor) ps = L.po_floor;
      if (L.po_cut != UNDEF && pe >= L.po_cut) pe = L.po_cut - 1;
      if (L.po_cut == 0) continue;
      prefix = pend_range_written(ps, pe, c, two, w2s, w2e, from, to);
    }
  }
  if (!prefix) {
    if (first_clause) { L.flags |= RGB_F_RESEND_PENDING; return 0; }
    return RGB_INV_WRITTEN_NOT_PREFIX;
  }
  if (pend_nonempty(L)) {
    if (c + 1 > L.pend) L.pend = c + 1;
    pend_canon(L);
  }
  pend_old_floor(L, c + 1);
  if (first_clause) {
    changed = !(L.lwi == c && L.lwt == term);
    L.lwi = c; L.lwt = term;
  }
  return 0;
}

/* ra_log:handle_event({snapshot_written,{Idx,Term},_,snapshot,_,_}) (src/ra_log.erl:1054-1150):
 * only when the range is defined and Idx >= its first index; last_written follows the snapshot
 * when it is not above it; the range is truncated behind the snapshot (ra_range:truncate/2).
 * The run table loses its leading runs right here (no other log edit can share the message). */
template <class Lane>
__device__ __forceinline__ bool log_snapshot_written(Lane &L, u64 idx, u64 term) {
  if (!(range_nonempty(L) && idx >= L.first)) return false;
  bool changed = false;
  if (!(L.lwi > idx)) {
    changed = !(L.lwi == idx && L.lwt == term);
    L.lwi = idx; L.lwt = term;
  }
  if (idx >= L.li) {
    L.li = idx; L.lt = term; L.first = idx + 1; L.n_runs = 0;       /* range undefined */
  } else {
    const u64 nf = idx + 1;
    const int k = find_run(L, nf);
    u64 *runs = const_cast<u64 *>(L.runs);
    if (k > 0) {
      /* runs k .. n_runs-1 move to the front.  The in-memory ones (k .. n_runs-3) go FOUR AT A TIME -- four independent
       * 16-byte loads, then four stores (a destination slot lies below every later source: k > 0) -- where round 4
       * moved them word by word, every load behind the previous store's acknowledgement: in a train a wavefront with
       * one lane that released ten runs spent 12 us here, and 3 in 4 wavefronts of the NEXT tick held a server that
       * waited for it (profiles/r05_train_timeline.txt).  The newest two runs are in registers. */
      ulonglong2 *rp = reinterpret_cast<ulonglong2 *>(runs);
      const int n_mem = (int)L.n_runs - 2 - k;          /* in-memory runs that stay */
#ifdef RGB_X_DECLINE_HIST
      atomicAdd(const_cast<u64 *>(L.dbg_hist) + 96 + (n_mem < 0 ? 0 : n_mem > 15 ? 15 : n_mem), 1ull);
#endif
#pragma unroll 1
      for (int i = 0; i < n_mem; i += 4) {
        ulonglong2 t0 = make_ulonglong2(0, 0), t1 = t0, t2 = t0, t3 = t0;
        t0 = ldg16(L.coh, rp + k + i);
        if (i + 1 < n_mem) t1 = ldg16(L.coh, rp + k + i + 1);
        if (i + 2 < n_mem) t2 = ldg16(L.coh, rp + k + i + 2);
        if (i + 3 < n_mem) t3 = ldg16(L.coh, rp + k + i + 3);
        ST16(rp + i, t0);
        if (i + 1 < n_mem) ST16(rp + i + 1, t1);
        if (i + 2 < n_mem) ST16(rp + i + 2, t2);
        if (i + 3 < n_mem) ST16(rp + i + 3, t3);
      }
      if ((int)L.n_runs - 2 >= k) ST16(rp + (L.n_runs - 2 - (unsigned)k), make_ulonglong2(L.prs, L.prt));
      ST16(rp + (L.n_runs - 1 - (unsigned)k), make_ulonglong2(L.lrs, L.lrt));
      L.n_runs -= (unsigned)k;
    }
    if (L.n_runs > 0) runs[0] = nf;
    if (L.lrs < nf) L.lrs = nf;
    if (L.n_runs == 2) L.prs = nf;                      /* run n-2 is now run 0 */
    L.first = nf;
  }
  L.si = idx; L.st = term;
  if (idx + 1 > L.pend) L.pend = idx + 1;     /* ra_seq:floor(SnapIdx+1, Pend0) :1100-1107, no live indexes */
  pend_old_floor(L, idx + 1);
  pend_canon(L);
  return changed;
}

/* ---- quorum ---- */

/* agreed_commit/1 (src/ra_server.erl:3684-3688) over up to 8 values held in registers:
 * descending order statistic nth = n/2+1 by rank counting (branch-free, no memory) */
template <int N>
__device__ __forceinline__ u64 agreed_commit(const u64 (&v)[N], const bool (&use)[N], int n) {
  /* the (n / 2 + 1)-th largest of the n used values.  Round 6: a SORT of the N candidates -- the unused ones as zeros,
   * which sort behind every used value or tie with used zeros -- by an odd-even transposition network, N (N - 1) / 2
   * compare-exchanges = as many 64-bit compares, then the pick of position n / 2; rounds 1-5 counted ranks (for every
   * value how many are greater / not smaller: 2 N^2 64-bit compares, 72 of the 110 in the reply fast path of groups of
   * five, the longest of the three bulk fast paths).  Same order statistic, value by value. */
  if (n <= 0) return UNDEF;
  u64 a[N];
#pragma unroll
  for (int i = 0; i < N; ++i) a[i] = use[i] ? v[i] : 0ull;
#pragma unroll
  for (int r = 0; r < N; ++r) {
#pragma unroll
    for (int i = r & 1; i + 1 < N; i += 2) {
      const bool sw = a[i + 1] > a[i];                    /* descending */
      const u64 hi = sw ? a[i + 1] : a[i], lo = sw ? a[i] : a[i + 1];
      a[i] = hi; a[i + 1] = lo;
    }
  }
  const int k = n / 2;                                    /* position of the (n / 2 + 1)-th largest */
  u64 res = a[0];
#pragma unroll
  for (int i = 1; i < N; ++i) res = (k == i) ? a[i] : res;
  return res;
}

/* match_indexes/1 :3671-3682, increment_commit_index/1 :3648-3657, evaluate_quorum/2 :3633-3646 */
template <int N, bool PL = false, class Lane>
__device__ __forceinline__ void evaluate_quorum(Lane &L) {
  u64 v[N + 1];
  bool use[N + 1];
  const unsigned self = self_of(L);
  int n = 1;
  v[N] = L.lwi; use[N] = true;                     /* the leader's last WRITTEN index */
#pragma unroll
  for (int i = 0; i < N; ++i) {
    bool u = ((unsigned)i != self) && present(L, i) && voter(L, i);
    v[i] = mi_get<N, PL>(L, i); use[i] = u;
    n += u ? 1 : 0;
  }
  const u64 ci0 = L.ci;
  u64 p = agreed_commit<N + 1>(v, use, n);
  u64 t = srv_fetch_term(L, p);
  if (t != UNDEF && t == L.ct) L.ci = p;           /* Raft 5.4.2; NO max() */
  if (L.ci > ci0) L.flags |= RGB_F_AUX_EVAL;
  if (apply_to(L, L.ci)) L.flags |= RGB_F_APPLIED;
}

/* make_pipelined_rpc_effects/3 :2285-2346 + make_rpc_effect/5 :2382-2416 +
 * make_append_entries_rpc/6 :2418-2435.  One pass: peer cursors change in registers and rpc
 * records go to this message's private slots, so when an assertion of the reference fails
 * (non-zero return) nothing has been committed: the decision reports n_rpcs = 0. */
template <int N, bool EMIT, bool PL = false, class Lane>
__device__ __forceinline__ int pipeline_rpcs(Lane &L, bool force, u32 max_pipe, u32 max_batch, bool &more,
                             unsigned &n_out, rgb_rpc *rpcs, u32 slot_base, u32 msg_index) {
  const unsigned self = self_of(L);
  const u64 next_log = next_log_index(L);
  more = false;
  n_out = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if ((unsigned)i == self || !present(L, i) || !status_normal(L, i)) continue;
    const u64 mi = mi_get<N, PL>(L, i), ni = ni_get<N, PL>(L, i), cis = cs_get<N, PL>(L, i);
    if (!(ni < next_log || cis < L.ci)) continue;
    long long inflight = (long long)(ni - mi) - 1;
    if (!(inflight < (long long)max_pipe || force)) continue;
    long long room = (long long)max_pipe - inflight;
    long long bs = (long long)max_batch < room ? (long long)max_batch : room;
    if (bs < 1) bs = 1;
    u64 prev = ni - 1;
    u64 prev_term = fetch_term(L, prev);
    u64 new_ni, rp_idx, rp_term;
    unsigned kind, n_ent = 0;
    if (prev_term == UNDEF && !(L.si != UNDEF && L.si == prev)) {
      /* next_index 0: PrevIdx is -1 in the reference's integers, below any snapshot index */
      if (L.si == UNDEF || !(ni == 0 || prev < L.si)) return RGB_INV_PIPELINE_PREV_UNDEFINED;
      kind = RGB_RPC_SNAPSHOT; rp_idx = L.si; rp_term = L.st; new_ni = L.si;
      if (EMIT) L.flags |= RGB_F_SEND_SNAPSHOT;
    } else {
      if (prev_term == UNDEF) prev_term = L.st;
      u64 to = prev + (u64)bs < L.li ? prev + (u64)bs : L.li;
      kind = RGB_RPC_AER; rp_idx = prev; rp_term = prev_term;
      n_ent = (unsigned)(to >= prev + 1 ? to - prev : 0);
      new_ni = to + 1;
    }
    if (!(new_ni >= ni)) return RGB_INV_NEXT_INDEX_REGRESSED;
    n_out += 1;
    long long new_inflight = (long long)(new_ni - mi) - 1;
    if (new_ni < next_log && new_inflight < (long long)max_pipe) more = true;
    if (EMIT) {
      ni_set<N, PL>(L, i, new_ni);
      /* commit_index_sent := commit_index.  The value is not kept per peer: nothing reads pcs[i] again in this
       * message and commit_index does not move after pipelining, so the commit step stores L.ci itself (ten
       * VGPRs fewer across the loop for N = 5) */
      L.dcs_ci |= 1u << i;
      /* fixed slot: this message's (n_out-1)-th record */
      emit_rpc(L, rpcs, slot_base, n_out - 1u, msg_index, (u64)(unsigned)i | ((u64)kind << 8) | ((u64)(n_ent & 0xFFFFu) << 16),
               rp_idx, rp_term, new_ni);
    }
  }
  return 0;
}

/* ------------------------------------------------------------------ elections ---- */

/* the leader branch of handle_candidate(#request_vote_result{vote_granted=true}) :1055-1058 with
 * initialise_peers/1 :3234-3242 */
template <int N, class Lane>
__device__ __forceinline__ void become_leader(Lane &L) {
  const u64 ni = next_log_index(L);
  load_peers<N>(L);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (!present(L, i)) continue;
    L.pmi[i] = 0; L.pni[i] = ni; L.pcs[i] = 0;
    L.dmi |= 1u << i; L.dni |= 1u << i; L.dcs |= 1u << i;
  }
  L.pk = pk_set(L.pk, PK_STATUS_SH, 8, 0xFF);
  L.pk = pk_set(L.pk, PK_BACKOFF_SH, 1, 0);
  set_leader_id(L, self_of(L));
  L.pk = pk_set(L.pk, PK_VOTES_SH, 4, 0);
  set_role(L, RGB_ROLE_LEADER);
  L.flags |= RGB_F_BECAME_LEADER;
}

/* one granted vote for a candidate in its own term (src/ra_server.erl:1045-1061) */
template <int N, class Lane>
__device__ __forceinline__ void candidate_vote_granted(Lane &L) {
  unsigned nv = (unsigned)pk_get(L.pk, PK_VOTES_SH, 4) + 1;
  if (nv == required_quorum(L)) become_leader<N>(L);
  else L.pk = pk_set(L.pk, PK_VOTES_SH, 4, nv);
}

template <class Lane>
__device__ __forceinline__ void vote_requests(Lane &L, u64 term, bool pre) {
  L.flags &= ~(u32)RGB_F_PRE_VOTE_REQS;
  L.flags |= RGB_F_SEND_VOTE_REQUESTS | (pre ? RGB_F_PRE_VOTE_REQS : 0u);
  L.has_reply = false;
  L.r_term = term; L.r_next = pre ? L.token : 0; L.r_last = L.li; L.r_lterm = L.lt;
  L.vote_reqs = true;
}

/* call_for_election(candidate,_) (src/ra_server.erl:2880-2899) + the self vote it casts */
template <int N, class Lane>
__device__ __forceinline__ void call_for_election_candidate(Lane &L) {
  const u64 new_term = L.ct + 1;
  update_term_and_voted_for(L, new_term, self_of(L));
  set_leader_id(L, SLOT_NONE4);
  L.pk = pk_set(L.pk, PK_VOTES_SH, 4, 0);
  set_role(L, RGB_ROLE_CANDIDATE);
  vote_requests(L, new_term, false);
  candidate_vote_granted<N>(L);
}

/* call_for_election(pre_vote,_) (src/ra_server.erl:2900-2924) + the self pre-vote (:1229-1246) */
template <int N, class Lane>
__device__ __forceinline__ void call_for_election_pre_vote(Lane &L, u64 token) {
  update_term_and_voted_for(L, L.ct, self_of(L));
  set_leader_id(L, SLOT_NONE4);
  L.pk = pk_set(L.pk, PK_VOTES_SH, 4, 0);
  L.token = token;
  set_role(L, RGB_ROLE_PRE_VOTE);
  vote_requests(L, L.ct, true);
  if (!pk_get(L.pk, PK_NONVOTER_SH, 1)) {
    unsigned nv = (unsigned)pk_get(L.pk, PK_VOTES_SH, 4) + 1;
    if (nv == required_quorum(L)) call_for_election_candidate<N>(L);
    else L.pk = pk_set(L.pk, PK_VOTES_SH, 4, nv);
  }
}

/* process_pre_vote/3 (src/ra_server.erl:2926-2983); the server stays in its role */
template <class Lane>
__device__ __forceinline__ int process_pre_vote(Lane &L) {
  const u64 token = L.c;
  if (L.term >= L.ct) {
    update_term(L, L.term);                 /* a pre-vote never sets voted_for */
    const bool up = (L.b > L.lt) || (L.b == L.lt && L.a >= L.li);
    const u32 theirs = L.n_entries, ours = (u32)(L.macver & 0xFFFFFFFFull), eff = (u32)(L.macver >> 32);
    if (up && L.gap > RGB_PROTO_VERSION) {
      pre_vote_reply(L, L.term, token, false, L.from);
    } else if (up && (theirs == eff || (theirs >= eff && theirs <= ours))) {
      pre_vote_reply(L, L.term, token, true, L.from);
    } else if (up) {
      pre_vote_reply(L, L.term, token, false, L.from);
      L.flags |= RGB_F_START_ELECTION_TIMEOUT;
    } else if (role_of(L) == RGB_ROLE_FOLLOWER) {
      L.flags |= RGB_F_START_ELECTION_TIMEOUT;             /* no reply, :2968-2969 */
    } else {
      pre_vote_reply(L, L.term, token, false, L.from);
    }
    return 0;
  }
  pre_vote_reply(L, L.ct, token, false, L.from);
  return 0;
}

/* make_all_rpcs/1 :2353-2367 -> make_rpcs_for/2 :2369-2377: one rpc (batch 1) per normal peer,
 * next_index NOT advanced */
template <int N, bool PL = false, class Lane>
__device__ __forceinline__ int make_all_rpcs(Lane &L, unsigned &n_out, rgb_rpc *rpcs, u32 slot_base,
                                             u32 msg_index, bool only_stale = false) {
  const unsigned self = self_of(L);
  update_heartbeat_rpc_effects<N>(L);                                /* :2354-2355 */
  if (!PL) load_peers<N>(L);
  n_out = 0;
  /* make_all_rpcs/1 keeps peers in {snapshot_backoff,_} as well and cancels their retry timers
   * (:2356-2363); stale_peers/1 (the tick) only takes normal ones */
  unsigned backoff = 0;
  if (!only_stale && pk_get(L.pk, PK_BACKOFF_SH, 1)) backoff = (unsigned)ldg8(L.coh, qry_row(L) + QRY_BACKOFF) & 0xFFu;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if ((unsigned)i == self || !present(L, i)) continue;
    if (!status_normal(L, i)) {
      if (!((backoff >> i) & 1u)) continue;
      L.cancel_mask |= 1u << i;
      L.flags |= RGB_F_CANCEL_SNAPSHOT_RETRY;
    }
    /* make_rpcs/1 on tick: stale_peers/1 :3012-3030 -- unconfirmed items or a newer commit index */
    const u64 ni_i = ni_get<N, PL>(L, i);
    if (only_stale && !(mi_get<N, PL>(L, i) + 1 < ni_i || cs_get<N, PL>(L, i) < L.ci)) continue;
    const u64 prev = ni_i - 1;
    u64 prev_term = fetch_term(L, prev);
    u64 rp_idx, rp_term, new_ni;
    unsigned kind, n_ent = 0;
    if (prev_term == UNDEF && !(L.si != UNDEF && L.si == prev)) {
      if (L.si == UNDEF || !(ni_i == 0 || prev < L.si)) return RGB_INV_PIPELINE_PREV_UNDEFINED;
      kind = RGB_RPC_SNAPSHOT; rp_idx = L.si; rp_term = L.st; new_ni = L.si;
      L.flags |= RGB_F_SEND_SNAPSHOT;
    } else {
      if (prev_term == UNDEF) prev_term = L.st;
      u64 to = prev + 1 < L.li ? prev + 1 : L.li;
      kind = RGB_RPC_AER; rp_idx = prev; rp_term = prev_term;
      n_ent = (unsigned)(to >= prev + 1 ? to - prev : 0);
      new_ni = to + 1;
    }
    n_out += 1;
    emit_rpc(L, rpcs, slot_base, n_out - 1u, msg_index, (u64)(unsigned)i | ((u64)kind << 8) | ((u64)(n_ent & 0xFFFFu) << 16),
             rp_idx, rp_term, new_ni);
  }
  return 0;
}

/* ------------------------------------------------------------------ follower ---- */

/* drop_existing/3 (src/ra_server.erl:3700-3708) run-wise: number of leading entries of the
 * message that already exist with the same term.  Equivalent to the per-entry
 * ra_log:exists loop because run tables are canonical (adjacent runs differ in term). */
template <class Lane>
__device__ __forceinline__ u32 drop_existing(const Lane &L) {
  if (L.n_entries == 0) return 0;
  const u64 base = L.a + 1 + (u64)L.gap;
  if (!range_nonempty(L) || base > L.li || base < L.first) return 0;
  u32 k = 0;
  /* segment 0: [base .. base+n_run0-1] term run0_term; segment 1: the rest, run1_term */
  for (int seg = 0; seg < 2; ++seg) {
    u32 cnt = seg == 0 ? L.n_run0 : L.n_entries - L.n_run0;
    if (seg == 0 && cnt > L.n_entries) cnt = L.n_entries;
    if (cnt == 0) continue;
    u64 s = base + k, e = s + cnt - 1;
    u64 t = seg == 0 ? L.run0_term : L.run1_term;
    if (s > L.li) return k;
    int r = find_run(L, s);
    if (r < 0) return k;
    const bool cached = Lane::walk_cache && r == L.wc_k;      /* find_run's walk just ended in it */
    u64 rterm = ((unsigned)r == L.n_runs - 1) ? L.lrt : ((unsigned)r == L.n_runs - 2) ? L.prt :
                cached ? L.wc_term : run_word(L, 2 * r + 1);
    if (rterm != t) return k;
    u64 rend = ((unsigned)r == L.n_runs - 1) ? L.li : ((unsigned)r == L.n_runs - 2) ? L.lrs - 1 :
               ((unsigned)r == L.n_runs - 3) ? L.prs - 1 : cached ? L.wc_next - 1 : run_word(L, 2 * (r + 1)) - 1;
    if (rend >= e) { k += cnt; continue; }
    k += (u32)(rend - s + 1);
    return k;
  }
  return k;
}

/* handle_follower(#append_entries_rpc{}) (src/ra_server.erl:1283-1440) */
template <class Lane>
__device__ __forceinline__ int follower_aer(Lane &L) {
  const u64 cur_term = L.ct;
  if (!(L.term >= cur_term)) {
    aer_reply(L, cur_term, false, L.from);                          /* :1431-1440 */
    return 0;
  }
  const u64 pli = L.a, plt = L.b, leader_commit = L.c;
  const u64 last_applied = L.la;
  set_leader_id(L, slot8to4(L.from));                               /* :1298 */
  update_term(L, L.term);
  int h = has_log_entry_or_snapshot(L, pli, plt);
  if (h == HLE_OK) {
    u32 k = drop_existing(L);
    u64 last_valid = k == 0 ? pli : (L.a + (u64)L.gap + k);
    if (k == L.n_entries) {
      /* nothing new to write, :1304-1364 */
      const u64 local_last = L.li;
      bool validated;
      if (L.n_entries == 0 && local_last > pli) {
        if (pli < last_applied) return RGB_INV_TRUNCATE_BELOW_APPLIED;
        int rc = log_set_last_index(L, pli);
        if (rc) return rc;
        L.flags |= RGB_F_TRUNCATED;
        validated = true;
      } else {
        validated = local_last <= last_valid;
      }
      if (validated) {
        L.ci = leader_commit;                                       /* not clamped, :1331 */
        L.flags |= RGB_F_LEADER_MSG;
        evaluate_commit_index_follower(L);
        aer_reply(L, L.term, true, L.from);
      } else {
        /* :1344-1363: success reply for what we have, term = PRE-update CurTerm */
        u64 v = last_applied > last_valid ? last_applied : last_valid;
        u64 vt = srv_fetch_term(L, v);
        L.has_reply = true;
        L.flags |= RGB_F_REPLY | RGB_F_REPLY_SUCCESS;
        L.r_term = cur_term; L.r_next = v + 1; L.r_last = v; L.r_lterm = vt;
        L.reply_to = L.from;
      }
      return 0;
    }
    /* :1365-1389 */
    const u64 fst = L.a + 1 + (u64)L.gap + k;
    L.ci = leader_commit;
    if (fst < last_applied) return RGB_INV_WRITE_BELOW_APPLIED;
    int rc = log_write(L, k);
    if (rc) return rc;
    L.flags |= RGB_F_WROTE | RGB_F_LEADER_MSG;
    L.w_first = fst; L.w_last = L.a + (u64)L.gap + L.n_entries;   /* what went to the WAL (the range may be gone) */
    evaluate_commit_index_follower(L);
    return 0;                                                       /* reply comes on written */
  }
  if (h == HLE_MISSING) {
    aer_reply(L, L.term, false, L.from);                            /* :1390-1404 */
    L.flags |= RGB_F_LEADER_MSG;
    set_role(L, RGB_ROLE_AWAIT_CONDITION);
    L.pk = pk_set(L.pk, PK_COND_SH, 2, RGB_COND_MISSING);
  } else {
    /* :1405-1429, mismatch_append_entries_reply/3 :3614-3622 */
    u64 lat = srv_fetch_term(L, last_applied);
    if (lat == UNDEF) return RGB_INV_MISMATCH_TERM_UNDEFINED;
    L.has_reply = true;
    L.flags |= RGB_F_REPLY | RGB_F_LEADER_MSG;
    L.r_term = L.term; L.r_next = last_applied + 1; L.r_last = last_applied; L.r_lterm = lat;
    L.reply_to = L.from;
    set_role(L, RGB_ROLE_AWAIT_CONDITION);
    L.pk = pk_set(L.pk, PK_COND_SH, 2, RGB_COND_TERM_MISMATCH);
  }
  L.cr0 = L.r_term; L.cr1 = L.r_next; L.cr2 = L.r_last; L.cr3 = L.r_lterm;
  L.cond_dirty = true;
  L.pk = pk_set(L.pk, PK_CONDLDR_SH, 4, slot8to4(L.from));
  return 0;
}

/* handle_follower(#request_vote_rpc{}) (src/ra_server.erl:1483-1529) */
template <class Lane>
__device__ __forceinline__ int follower_request_vote(Lane &L) {
  if (pk_get(L.pk, PK_NONVOTER_SH, 1)) return 0;
  const unsigned cand4 = slot8to4(L.from);
  const unsigned voted = (unsigned)pk_get(L.pk, PK_VOTED_SH, 4);
  if (L.term == L.ct && voted != SLOT_NONE4 && voted != cand4) {
    vote_reply(L, L.term, false, L.from);
    return 0;
  }
  if (L.term >= L.ct) {
    update_term(L, L.term);
    /* is_candidate_log_up_to_date/3 :3157-3166 */
    bool up = (L.b > L.lt) || (L.b == L.lt && L.a >= L.li);
    if (up) {
      update_term_and_voted_for(L, L.term, cand4);
      vote_reply(L, L.term, true, L.from);
    } else {
      vote_reply(L, L.term, false, L.from);
    }
    return 0;
  }
  vote_reply(L, L.ct, false, L.from);
  return 0;
}

template <int N, class Lane>
__device__ __forceinline__ int handle_follower(Lane &L) {
  switch (L.kind) {
    case RGB_MSG_AER:          return follower_aer(L);
    case RGB_MSG_REQUEST_VOTE: return follower_request_vote(L);
    case RGB_MSG_WRITTEN: {
      /* :1457-1474 */
      bool changed;
      int rc = log_written(L, L.term, L.a, L.b, changed);
      if (rc) return rc;
      unsigned l4 = (unsigned)pk_get(L.pk, PK_LEADER_SH, 4);
      if (changed && l4 != SLOT_NONE4) aer_reply(L, L.ct, true, slot4to8(l4));
      return 0;
    }
    case RGB_MSG_SNAPSHOT_WRITTEN: {
      bool changed = log_snapshot_written(L, L.a, L.b);
      unsigned l4 = (unsigned)pk_get(L.pk, PK_LEADER_SH, 4);
      if (changed && l4 != SLOT_NONE4) aer_reply(L, L.ct, true, slot4to8(l4));
      return 0;
    }
    case RGB_MSG_AER_REPLY:    update_term(L, L.term); return 0;     /* :1530-1533 */
    case RGB_MSG_VOTE_RESULT:  return 0;                             /* :1609-1611 */
    case RGB_MSG_HEARTBEAT_RPC:
      if (L.term >= L.ct) {                                          /* :1441-1450 */
        update_term(L, L.term);
        set_leader_id(L, slot8to4(L.from));
        heartbeat_reply(L, L.term, L.a, L.from);
      } else {
        heartbeat_reply(L, L.ct, L.a, L.from);                       /* :1451-1456 */
      }
      return 0;
    case RGB_MSG_HEARTBEAT_REPLY: update_term(L, L.term); return 0;  /* :1534-1537 */
    case RGB_MSG_PRE_VOTE_RESULT: return 0;                          /* :1612-1614 */
    case RGB_MSG_PRE_VOTE_RPC:
      if (pk_get(L.pk, PK_NONVOTER_SH, 1)) return 0;                 /* :1475-1480 */
      return process_pre_vote(L);                                    /* :1481-1482 */
    case RGB_MSG_ELECTION_TIMEOUT:
      if (pk_get(L.pk, PK_NONVOTER_SH, 1)) return 0;                 /* :1619-1624 */
      call_for_election_pre_vote<N>(L, L.c);                         /* :1625-1626 */
      return 0;
    default: L.flags |= RGB_F_UNHANDLED; return 0;
  }
}

/* -------------------------------------------------------------------- leader ---- */
template <int N, bool PL = false, class Lane>
__device__ __forceinline__ int handle_leader(Lane &L, bool &reprocess, const rgb_dev &dev, rgb_rpc *rpcs,
                             u32 slot_base, u32 msg_index, unsigned &n_rpcs) {
  switch (L.kind) {
    case RGB_MSG_AER_REPLY: {
      const unsigned peer = L.from;
      const bool success = (L.mflags & RGB_MF_SUCCESS) != 0;
      if (success && L.term == L.ct) {
        /* :532-571 */
        if (!present(L, peer)) return 0;
        if (!PL) load_peers<N>(L);
        if (L.b > mi_of<N, PL>(L, peer)) mi_put<N, PL>(L, peer, L.b);
        if (L.a > ni_of<N, PL>(L, peer)) ni_put<N, PL>(L, peer, L.a);
        evaluate_quorum<N, PL>(L);
        L.flags |= RGB_F_PIPELINE;
        return 0;
      }
      if (L.term > L.ct) {
        /* :572-586 */
        if (!present(L, peer)) return 0;
        set_leader_id(L, SLOT_NONE4);
        update_term(L, L.term);
        set_role(L, RGB_ROLE_FOLLOWER);
        return 0;
      }
      if (!success) {
        /* :587-652 */
        if (!present(L, peer)) return 0;
        if (!PL) load_peers<N>(L);
        u64 mi = mi_of<N, PL>(L, peer), ni = ni_of<N, PL>(L, peer);
        const u64 peer_next = L.a, peer_last = L.b, peer_last_term = L.c;
        u64 t = fetch_term(L, peer_last);
        if (t == UNDEF) {
          ni = peer_next;
        } else if (t == peer_last_term && peer_last >= mi) {
          mi = peer_last; ni = peer_next;
        } else if (peer_last < mi) {
          mi = peer_last; ni = peer_last + 1;
        } else {
          long long x = (long long)ni - 1, y = (long long)peer_last;
          long long mn = x < y ? x : y;
          long long lo = (long long)mi + 1;
          ni = (u64)(mn > lo ? mn : lo);
        }
        /* registers only: nothing reaches memory unless the message commits */
        mi_put<N, PL>(L, peer, mi);
        ni_put<N, PL>(L, peer, ni);
        bool more; unsigned cnt;
        int rc = pipeline_rpcs<N, true, PL>(L, false, dev.max_pipeline_count, dev.max_aer_batch, more, cnt,
                                        rpcs, slot_base, msg_index);
        if (rc) return rc;
        n_rpcs = cnt;
        return 0;
      }
      L.flags |= RGB_F_UNHANDLED;                                    /* :1038-1040 */
      return 0;
    }
    case RGB_MSG_AER: {
      if (L.term > L.ct) {
        set_leader_id(L, SLOT_NONE4);                                /* :835-844 */
        update_term(L, L.term);
        set_role(L, RGB_ROLE_FOLLOWER);
        reprocess = true;
        return 0;
      }
      if (L.term == L.ct) return RGB_INV_LEADER_SAW_AER_SAME_TERM;   /* :845-849 */
      aer_reply(L, L.ct, false, L.from);                             /* :850-854 */
      return 0;
    }
    case RGB_MSG_REQUEST_VOTE: {
      if (L.term > L.ct) {
        if (!present(L, L.from)) return 0;                           /* :928-942 */
        set_leader_id(L, SLOT_NONE4);
        update_term(L, L.term);
        set_role(L, RGB_ROLE_FOLLOWER);
        reprocess = true;
        return 0;
      }
      vote_reply(L, L.ct, false, L.from);                            /* :943-945 */
      return 0;
    }
    case RGB_MSG_WRITTEN: {
      bool changed;
      int rc = log_written(L, L.term, L.a, L.b, changed);            /* :739-744 */
      if (rc) return rc;
      if (!PL) load_peers<N>(L);
      evaluate_quorum<N, PL>(L);
      L.flags |= RGB_F_PIPELINE;
      return 0;
    }
    case RGB_MSG_PIPELINE_RPCS:
    case RGB_MSG_APPEND: {
      if (L.kind == RGB_MSG_PIPELINE_RPCS && (L.mflags & RGB_MF_TICK)) {
        unsigned cnt;                                                /* tick_timeout: make_rpcs/1 :2348-2351 */
        int rc = make_all_rpcs<N, PL>(L, cnt, rpcs, slot_base, msg_index, true);
        if (rc) return rc;
        n_rpcs = cnt;
        return 0;
      }
      bool force = false;
      if (!PL) load_peers<N>(L);
      if (L.kind == RGB_MSG_APPEND) {
        /* {command,_} :653-693 / {commands,_} :695-738: ra_log:append of n entries at
         * next_index in the current term (registers only until commit) */
        force = (L.mflags & RGB_MF_FORCE) != 0;
        if (L.n_entries > 0) {
          u64 nidx = next_log_index(L);
          if (!range_nonempty(L)) { L.first = nidx; L.n_runs = 0; }
          push_segment(L, nidx, L.ct);
          L.li = nidx + (L.n_entries - 1);
          L.lt = L.ct;
          if (nidx < L.pend) L.pend = nidx;   /* ra_seq:limit(Idx-1) + append(Idx) :503-505 */
          pend_old_limit(L, nidx);
        }
      }
      bool more; unsigned cnt;
      int rc = pipeline_rpcs<N, true, PL>(L, force, dev.max_pipeline_count, dev.max_aer_batch, more, cnt,
                                          rpcs, slot_base, msg_index);
      if (rc) return rc;
      n_rpcs = cnt;
      if (L.kind == RGB_MSG_PIPELINE_RPCS && more) L.flags |= RGB_F_PIPELINE;   /* :793-801 */
      return 0;
    }
    case RGB_MSG_PRE_VOTE_RPC: {
      if (L.term > L.ct) {
        if (!present(L, L.from)) return 0;                           /* :946-960 */
        set_leader_id(L, SLOT_NONE4);
        update_term(L, L.term);
        set_role(L, RGB_ROLE_FOLLOWER);
        reprocess = true;
        return 0;
      }
      unsigned cnt;
      int rc = make_all_rpcs<N, PL>(L, cnt, rpcs, slot_base, msg_index);  /* :961-966 */
      if (rc) return rc;
      n_rpcs = cnt;
      return 0;
    }
    case RGB_MSG_CONSISTENT_QUERY:
      /* :855-860 / :868-873 with cluster_change_permitted = true (the host holds queries while
       * it is false, :861-867) */
      make_heartbeat_rpc_effects<N>(L);
      return 0;
    case RGB_MSG_HEARTBEAT_RPC:
      if (L.term > L.ct) {                                           /* :880-889 */
        set_leader_id(L, SLOT_NONE4);
        update_term(L, L.term);
        set_role(L, RGB_ROLE_FOLLOWER);
        reprocess = true;
        return 0;
      }
      if (L.ct > L.term) { heartbeat_reply(L, L.ct, L.a, L.from); return 0; }   /* :890-897 */
      return RGB_INV_LEADER_SAW_HEARTBEAT_SAME_TERM;                 /* :898-903 */
    case RGB_MSG_HEARTBEAT_REPLY:                                    /* :904-927 */
      if (L.term == L.ct) {
        heartbeat_rpc_quorum<N>(L, L.a, L.from);
      } else if (L.term > L.ct) {
        set_leader_id(L, SLOT_NONE4);
        update_term(L, L.term);
        set_role(L, RGB_ROLE_FOLLOWER);
      }
      return 0;
    case RGB_MSG_VOTE_RESULT:                                        /* :967-969 */
    case RGB_MSG_PRE_VOTE_RESULT:                                    /* :970-972 */
      return 0;
    case RGB_MSG_SNAPSHOT_WRITTEN:                                   /* :745-747 */
      log_snapshot_written(L, L.a, L.b);
      return 0;
    default: L.flags |= RGB_F_UNHANDLED; return 0;
  }
}

/* ----------------------------------------------------------------- candidate ---- */
template <int N, class Lane>
__device__ __forceinline__ int handle_candidate(Lane &L, bool &reprocess) {
  switch (L.kind) {
    case RGB_MSG_VOTE_RESULT: {
      const bool granted = (L.mflags & RGB_MF_SUCCESS) != 0;
      if (granted && L.term == L.ct) {
        candidate_vote_granted<N>(L);                                /* :1045-1061 */
        return 0;
      }
      if (L.term > L.ct) {
        update_term_and_voted_for(L, L.term, SLOT_NONE4);            /* :1062-1069 */
        set_role(L, RGB_ROLE_FOLLOWER);
      }
      return 0;
    }
    case RGB_MSG_AER:
      if (L.term >= L.ct) {
        update_term_and_voted_for(L, L.term, SLOT_NONE4);            /* :1072-1075 */
        set_role(L, RGB_ROLE_FOLLOWER);
        reprocess = true;
        return 0;
      }
      aer_reply(L, L.ct, false, L.from);                             /* :1076-1080 */
      return 0;
    case RGB_MSG_AER_REPLY:
      if (L.term > L.ct) {
        update_term_and_voted_for(L, L.term, SLOT_NONE4);            /* :1098-1106 */
        set_role(L, RGB_ROLE_FOLLOWER);
        return 0;
      }
      L.flags |= RGB_F_UNHANDLED;
      return 0;
    case RGB_MSG_REQUEST_VOTE:
      if (L.term > L.ct) {
        update_term_and_voted_for(L, L.term, SLOT_NONE4);            /* :1107-1114 */
        set_role(L, RGB_ROLE_FOLLOWER);
        reprocess = true;
        return 0;
      }
      vote_reply(L, L.ct, false, L.from);                            /* :1123-1125 */
      return 0;
    case RGB_MSG_WRITTEN:
      { bool changed; return log_written(L, L.term, L.a, L.b, changed); }   /* :1157-1160 */
    case RGB_MSG_PRE_VOTE_RPC:
      if (L.term > L.ct) {
        update_term_and_voted_for(L, L.term, SLOT_NONE4);            /* :1116-1122 */
        set_role(L, RGB_ROLE_FOLLOWER);
        reprocess = true;
        return 0;
      }
      return process_pre_vote(L);                                    /* :1127-1131 */
    case RGB_MSG_HEARTBEAT_RPC:
      if (L.term >= L.ct) {                                          /* :1081-1084 */
        update_term_and_voted_for(L, L.term, SLOT_NONE4);
        set_role(L, RGB_ROLE_FOLLOWER);
        reprocess = true;
        return 0;
      }
      heartbeat_reply(L, L.ct, L.a, L.from);                         /* :1085-1090 */
      return 0;
    case RGB_MSG_HEARTBEAT_REPLY:
      if (L.term > L.ct) {                                           /* :1091-1099 */
        update_term_and_voted_for(L, L.term, SLOT_NONE4);
        set_role(L, RGB_ROLE_FOLLOWER);
        return 0;
      }
      L.flags |= RGB_F_UNHANDLED;                                    /* catch-all */
      return 0;
    case RGB_MSG_PRE_VOTE_RESULT: return 0;                          /* :1135-1137 */
    case RGB_MSG_SNAPSHOT_WRITTEN:
      log_snapshot_written(L, L.a, L.b);                             /* :1157-1160 */
      return 0;
    case RGB_MSG_ELECTION_TIMEOUT:
      call_for_election_candidate<N>(L);                             /* :1161-1162 */
      return 0;
    default: L.flags |= RGB_F_UNHANDLED; return 0;
  }
}

/* ------------------------------------------------------------------ pre_vote ---- */
template <int N, class Lane>
__device__ __forceinline__ int handle_pre_vote(Lane &L, bool &reprocess) {
  switch (L.kind) {
    case RGB_MSG_AER:
      if (L.term >= L.ct) {
        update_term(L, L.term);                                      /* :1192-1197 */
        L.pk = pk_set(L.pk, PK_VOTES_SH, 4, 0);
        set_role(L, RGB_ROLE_FOLLOWER);
        reprocess = true;
        return 0;
      }
      L.flags |= RGB_F_UNHANDLED;
      return 0;
    case RGB_MSG_REQUEST_VOTE:
      if (L.term > L.ct) {
        update_term(L, L.term);                                      /* :1214-1219 */
        L.pk = pk_set(L.pk, PK_VOTES_SH, 4, 0);
        set_role(L, RGB_ROLE_FOLLOWER);
        reprocess = true;
        return 0;
      }
      L.flags |= RGB_F_UNHANDLED;
      return 0;
    case RGB_MSG_VOTE_RESULT: return 0;                              /* :1249-1251 */
    case RGB_MSG_WRITTEN:
      { bool changed; return log_written(L, L.term, L.a, L.b, changed); }   /* :1257-1260 */
    case RGB_MSG_PRE_VOTE_RESULT: {
      const bool granted = (L.mflags & RGB_MF_SUCCESS) != 0;
      if (L.term > L.ct) {
        update_term(L, L.term);                                      /* :1219-1228 */
        L.pk = pk_set(L.pk, PK_VOTES_SH, 4, 0);
        set_role(L, RGB_ROLE_FOLLOWER);
        return 0;
      }
      if (granted && L.term == L.ct && L.c == L.token && !pk_get(L.pk, PK_NONVOTER_SH, 1)) {
        unsigned nv = (unsigned)pk_get(L.pk, PK_VOTES_SH, 4) + 1;    /* :1229-1246 */
        if (nv == required_quorum(L)) call_for_election_candidate<N>(L);
        else L.pk = pk_set(L.pk, PK_VOTES_SH, 4, nv);
      }
      return 0;
    }
    case RGB_MSG_SNAPSHOT_WRITTEN:
      log_snapshot_written(L, L.a, L.b);                             /* :1257-1260 */
      return 0;
    case RGB_MSG_HEARTBEAT_RPC:
      if (L.term >= L.ct) {                                          /* :1198-1203 */
        update_term(L, L.term);
        L.pk = pk_set(L.pk, PK_VOTES_SH, 4, 0);
        set_role(L, RGB_ROLE_FOLLOWER);
        reprocess = true;
        return 0;
      }
      heartbeat_reply(L, L.ct, L.a, L.from);                         /* :1204-1208 */
      return 0;
    case RGB_MSG_HEARTBEAT_REPLY:
      if (L.term > L.ct) {                                           /* :1209-1212 */
        update_term(L, L.term);
        L.pk = pk_set(L.pk, PK_VOTES_SH, 4, 0);
        set_role(L, RGB_ROLE_FOLLOWER);
        return 0;
      }
      L.flags |= RGB_F_UNHANDLED;                                    /* catch-all */
      return 0;
    case RGB_MSG_PRE_VOTE_RPC: return process_pre_vote(L);           /* :1250-1251 */
    case RGB_MSG_ELECTION_TIMEOUT:
      call_for_election_pre_vote<N>(L, L.c);                         /* :1255-1256 */
      return 0;
    default: L.flags |= RGB_F_UNHANDLED; return 0;
  }
}

/* ----------------------------------------------------------- await_condition ---- */
template <int N, class Lane>
__device__ __forceinline__ int handle_await_condition(Lane &L, bool &reprocess, const u64 *cond_row) {
  /* wal_down_condition/2 :2232-2233: the predicate is ra_log:can_write/1, which the host knows and passes along.
   * The follower's condition (:1377-1385) has no transition_to and no timeout map (follower, no effects); the leader's
   * (:660-668, PK_CONDTO) goes back to leader, on a timeout with [{next_event, cast, {transfer_leadership, Peer}}] */
  const bool wal_down = pk_get(L.pk, PK_COND_SH, 2) == RGB_COND_WAL_DOWN;
  const bool can_write = (L.mflags & RGB_MF_CAN_WRITE) != 0;
  const unsigned back = pk_get(L.pk, PK_CONDTO_SH, 1) ? RGB_ROLE_LEADER : RGB_ROLE_FOLLOWER;
  switch (L.kind) {
    case RGB_MSG_REQUEST_VOTE:
      set_role(L, RGB_ROLE_FOLLOWER);                                /* :1918-1919 */
      reprocess = true;
      return 0;
    case RGB_MSG_AWAIT_TIMEOUT: {
      if (wal_down) {                    /* :1932-1945: transition_to either way; the leader's timeout map has an effect */
        if (back == RGB_ROLE_LEADER && !can_write &&
            (pk_get(L.pk, PK_PRESENT_SH, 8) & ~(1ull << self_of(L))) != 0ull)
          L.flags |= RGB_F_TRANSFER_LEADERSHIP;
        set_role(L, back);
        return 0;
      }
      /* :1932-1945: predicate false -> stored effects, back to follower */
      L.has_reply = true;
      L.flags |= RGB_F_REPLY | RGB_F_LEADER_MSG;
      L.r_term = ldg8(L.coh, cond_row); L.r_next = ldg8(L.coh, cond_row + 1);
      L.r_last = ldg8(L.coh, cond_row + 2); L.r_lterm = ldg8(L.coh, cond_row + 3);
      L.reply_to = slot4to8((unsigned)pk_get(L.pk, PK_CONDLDR_SH, 4));
      set_role(L, RGB_ROLE_FOLLOWER);
      return 0;
    }
    case RGB_MSG_WRITTEN:
      { bool changed; return log_written(L, L.term, L.a, L.b, changed); }   /* :1946-1949 */
    case RGB_MSG_SNAPSHOT_WRITTEN:
      log_snapshot_written(L, L.a, L.b);                             /* :1946-1949 */
      return 0;
    case RGB_MSG_PRE_VOTE_RPC: return process_pre_vote(L);           /* :1920-1921 */
    case RGB_MSG_ELECTION_TIMEOUT:
      if (pk_get(L.pk, PK_NONVOTER_SH, 1)) return 0;                 /* :1922-1929 */
      call_for_election_pre_vote<N>(L, L.c);                         /* :1930-1931 */
      return 0;
    case RGB_MSG_AER: {
      /* follower_catchup_cond/3 :2201-2218 */
      bool pred = false;
      if (wal_down) {
        pred = can_write;
      } else if (L.term >= L.ct) {
        int h = has_log_entry_or_snapshot(L, L.a, L.b);
        if (h == HLE_OK) pred = true;
        else if (h == HLE_MISMATCH) pred = pk_get(L.pk, PK_COND_SH, 2) == RGB_COND_MISSING;
      }
      if (pred) { set_role(L, wal_down ? back : (unsigned)RGB_ROLE_FOLLOWER); reprocess = true; } /* :1950-1955 */
      return 0;
    }
    default:
      /* the catch-all clause :1950-1959: follower_catchup_cond/3 is false for anything but an append_entries_rpc;
       * the wal_down predicate does not look at the message */
      if (wal_down && can_write) { set_role(L, back); reprocess = true; }
      return 0;
  }
}

/* the 64-byte decision as 8 words (layout of rgb_decision) */
struct Dec { u64 w[8]; };

__device__ __forceinline__ void make_decision(Dec &d, u32 server, unsigned role, unsigned reply_to,
                                              unsigned n_rpcs, unsigned kind, u32 flags, u32 inv, u64 w2,
                                              u64 w3, u64 w4, u64 w5, u64 ci, u64 la, unsigned hb_mask = 0,
                                              unsigned cancel_mask = 0) {
  d.w[0] = (u64)server | ((u64)(role & 0xFF) << 32) | ((u64)(reply_to & 0xFF) << 40) |
           ((u64)(n_rpcs & 0xFF) << 48) | ((u64)(kind & 0xFF) << 56);
  d.w[1] = (u64)flags | ((u64)(inv & 0xFFFFu) << 32) | ((u64)(hb_mask & 0xFFu) << 48) |
           ((u64)(cancel_mask & 0xFFu) << 56);
  d.w[2] = w2; d.w[3] = w3; d.w[4] = w4; d.w[5] = w5; d.w[6] = ci; d.w[7] = la;
}

/* The compact form of a decision (include/ra_gpu_batch.h, "Compact decisions"): applied where a decision is staged for
 * its store, whatever path computed it; returns whether the record became 32 bytes.  Three shapes, each tested value
 * by value -- what does not fit stays a full record. */
#ifndef RGB_X_COMPACT
#define RGB_X_COMPACT 1
#endif
__device__ __forceinline__ bool compact_decision(Dec &d) {
  if (!RGB_X_COMPACT) return false;
  const u32 flags = (u32)d.w[1];
  const unsigned kind = (unsigned)(d.w[0] >> 56);
  if ((d.w[1] >> 32) != 0ull || (flags & RGB_F_COMPACT)) return false;             /* invariant / masks in use */
  const u32 plain = RGB_F_LEADER_MSG | RGB_F_APPLIED | RGB_F_AUX_EVAL | RGB_F_PIPELINE;     /* flags that carry no words */
  u64 A, B; u32 aux = 0;
  if ((flags & ~plain) == 0u && (kind == RGB_MSG_AER_REPLY || kind == RGB_MSG_WRITTEN)) {
    if ((d.w[2] | d.w[3] | d.w[4] | d.w[5]) != 0ull) return false;
    A = d.w[6]; B = d.w[7];                                                         /* counted */
  } else if ((flags & ~plain) == RGB_F_WROTE && kind == RGB_MSG_AER) {
    const u64 fst = d.w[3], lst = d.w[4], la = d.w[7];
    if ((d.w[2] | d.w[5]) != 0ull || fst > lst || lst - fst > 0xFFFFull || la > lst || lst - la > 0xFFFFull) return false;
    A = lst; B = d.w[6];
    aux = (u32)(lst - fst) | ((u32)(lst - la) << 16);                               /* wrote */
  } else if ((flags & ~plain) == (RGB_F_REPLY | RGB_F_REPLY_SUCCESS) && (kind == RGB_MSG_AER || kind == RGB_MSG_WRITTEN)) {
    const u64 term = d.w[2], next = d.w[3], last = d.w[4], lterm = d.w[5], ci = d.w[6], la = d.w[7];
    if (next == 0ull) return false;
    A = next - 1ull; B = term;
    const u64 d1 = A - last, d2 = term - lterm, d3 = ci + 512ull - A, d4 = A + 1ull - la;
    if (last > A || d1 > 0xFFull || lterm > term || d2 > 0xFull || ci + 512ull < A || d3 > 0x3FFull || la > A + 1ull ||
        d4 > 0x3FFull)
      return false;
    aux = (u32)d1 | ((u32)d2 << 8) | ((u32)d3 << 12) | ((u32)d4 << 22);            /* confirmed */
  } else {
    return false;
  }
  d.w[1] = (u64)(flags | RGB_F_COMPACT) | ((u64)aux << 32);
  d.w[2] = A; d.w[3] = B;
  return true;
}

/* One message against one server: everything between "message words in registers" and
 * "decision words in registers".  State loads/stores go straight to the server's lines. */
/* TR = the multi-tick train launch: state loads bypass the L1 (LaneT<true>::coh, see ldg8). */
/* SEQXP: written events of more than two ranges (RGB_MF_SEQX) are understood -- the kind-generic kernel, and the
 * written-only kernel rgb_submit_seq's batches run their written class through (rgb_tick_kernel<N, WRITTEN, true>) */
template <int N, int KIND, bool PRE = false, bool TR = false, bool SEQXP = (KIND < 0)>
__device__ __forceinline__ void process_message(const rgb_dev &dev, const ulonglong2 m0, const ulonglong2 m1,
                                                const ulonglong2 m2, const ulonglong2 m3, u32 i,
                                                rgb_rpc *__restrict__ rpcs, u32 rpc_slot_base,
                                                u32 msg_index_base, Dec &out, u64 *t_loaded = nullptr,
                                                const ulonglong2 *pre = nullptr, unsigned swz = 0,
                                                const ulonglong2 *prepeers = nullptr,
                                                const ulonglong2 *preruns = nullptr, ulonglong2 *rpc_stash = nullptr) {
  LaneT<TR, (N >= 6 && KIND == RGB_MSG_AER), SEQXP> L;
  L.rpc_stash = rpc_stash;   /* (the walk cache costs seven registers: only where it pays) */
  L.wc_k = -1; L.wc_start = L.wc_term = L.wc_next = 0;
  L.server = (u32)(m0.x & 0xFFFFFFFFull);
  /* KIND >= 0: compile-time message kind -- the clause switches fold and only that kind's path
   * (and its registers) remain */
  const unsigned wire_kind = (unsigned)((m0.x >> 32) & 0xFF);
  L.kind = KIND >= 0 ? (unsigned)KIND : wire_kind;
  L.from = (unsigned)((m0.x >> 40) & 0xFF);
  L.mflags = (unsigned)((m0.x >> 48) & 0xFF);
  L.gap = (unsigned)((m0.x >> 56) & 0xFF);
  L.term = m0.y; L.a = m1.x; L.b = m1.y; L.c = m2.x;
  L.n_entries = (u32)(m2.y & 0xFFFFFFFFull);
  L.n_run0 = (u32)(m2.y >> 32);
  L.run0_term = m3.x; L.run1_term = m3.y;
  if (L.n_run0 > L.n_entries) L.n_run0 = L.n_entries;

  if (L.kind == RGB_MSG_NOP) {
    make_decision(out, L.server, 0, RGB_NONE, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
    return;
  }
  if (L.server >= dev.n_servers || wire_kind != L.kind) {
    make_decision(out, L.server, 0xFF, RGB_NONE, 0, wire_kind, RGB_F_UNHANDLED, 0, 0, 0, 0, 0, 0, 0);
    return;
  }
  /* hot line: 8 x 16 B (15 live words) */
  u64 *hot = dev.hot + (size_t)L.server * RGB_HOT_WORDS;
  const ulonglong2 *hp = reinterpret_cast<const ulonglong2 *>(hot);
  ulonglong2 h0, h1, h2, h3, h4, h5, h6, h7;
  if (RGB_KNOB(dev, 8u)) { h0 = h1 = h2 = h3 = h4 = h5 = h6 = h7 = make_ulonglong2(0, 0); h0.y = 0x1Full << PK_PRESENT_SH; }
  else if (PRE) {     /* the wavefront fetched the lines cooperatively into LDS: pre = this lane's row */
    /* unpadded 128-byte rows: piece p sits at position p ^ swz (conflict-free 16-byte LDS reads, see the fetch) */
    /* h0 (term, packed) h1 (commit, applied) h2 (last index, term) h3 (last written) h4 (snapshot) h5 (first, last-run
     * start) h6 (last-run term, prev-run start) h7 (prev-run term, pending): by meaning, wherever the piece sits */
    h0 = pre[HOT_P_TERM ^ swz]; h1 = pre[HOT_P_CI ^ swz]; h2 = pre[HOT_P_LI ^ swz]; h3 = pre[HOT_P_LW ^ swz];
    h4 = pre[HOT_P_SI ^ swz]; h5 = pre[HOT_P_FIRST ^ swz]; h6 = pre[HOT_P_LRT ^ swz]; h7 = pre[HOT_P_PEND ^ swz];
  } else {
    h0 = ldg16(TR, hp + HOT_P_TERM); h1 = ldg16(TR, hp + HOT_P_CI); h2 = ldg16(TR, hp + HOT_P_LI); h3 = ldg16(TR, hp + HOT_P_LW);
    h4 = ldg16(TR, hp + HOT_P_SI); h5 = ldg16(TR, hp + HOT_P_FIRST); h6 = ldg16(TR, hp + HOT_P_LRT); h7 = ldg16(TR, hp + HOT_P_PEND);
  }
#ifdef RGB_PROFILE
  L.prof_noprobe = RGB_KNOB(dev, 32u); L.prof_nloads = 0;
#endif
#ifdef RGB_X_DECLINE_HIST
  L.dbg_hist = dev.dbg_buf;
#endif
  if (decltype(L)::seqx_ok) { L.seqx = dev.seq_ranges; L.n_seqx = dev.n_seq_ranges; } else { L.seqx = nullptr; L.n_seqx = 0; }
  L.qry_base = dev.qry;
  L.q_loaded = false; L.q_dirty = 0; L.hb_mask = 0;
  L.qself = 0; L.hb_term = L.hb_qi = L.q_consensus = 0; L.cancel_mask = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) L.qp[i] = 0;
  L.runs = dev.runs + (size_t)L.server * dev.max_runs * 2;
  L.peers = dev.peers + (size_t)L.server * dev.peer_stride;
  L.max_runs = dev.max_runs;
  L.peers_loaded = false; L.dmi = L.dni = L.dcs = 0; L.dcs_ci = 0;
  /* leader-side kinds: fetch the peers row in the same round trip as the hot line (the address
   * only depends on the message); kinds that need it rarely load it lazily */
  L.peers_lds = prepeers; L.peers_swz = swz; L.pdirty = 0;
  L.runs_lds = preruns;
  /* PL: the class kernel fetched this leader-side message's peers row into LDS with the hot row (128-byte rows:
   * 3..5 members): the clause code reads and writes it there, no register copy */
#ifndef RGB_X_COMMIT_WAIT
#define RGB_X_COMMIT_WAIT 1
#endif
#ifndef RGB_X_PL
#define RGB_X_PL 0     /* measured 1 % slower than the register arrays on the aged stream (23.6 vs 23.4 us), although it
                          removes every spill of the leader-side classes: kept as an option */
#endif
#ifndef RGB_X_PL_TICK
#define RGB_X_PL_TICK 1   /* the per-tick class kernel (4 wavefronts per SIMD, 128 registers): the peers row stays in LDS and
                             the commit re-reads the hot row from LDS (RGB_X_REREAD_TICK) -- no spills, where the register
                             arrays spilled 34 VGPRs (40 bytes of scratch per lane) */
#endif
#ifndef RGB_X_REREAD_TICK
#define RGB_X_REREAD_TICK 1
#endif
  constexpr bool PL = (TR ? RGB_X_PL : RGB_X_PL_TICK) && PRE && rgb_class_slice(1, (unsigned)N) == 32u &&
                      (KIND == RGB_MSG_AER_REPLY || KIND == RGB_MSG_APPEND || KIND == RGB_MSG_PIPELINE_RPCS);
  if (!PL && (L.kind == RGB_MSG_AER_REPLY || L.kind == RGB_MSG_APPEND || L.kind == RGB_MSG_PIPELINE_RPCS) &&
      !RGB_KNOB(dev, 64u)) {
    load_peers<N>(L);      /* from memory (the class kernel touched the row's line with the hot-line fetch) */
  }
#ifdef RGB_PROFILE
  if (RGB_KNOB(dev, 64u)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { L.pmi[i] = 0; L.pni[i] = 1; L.pcs[i] = 0; }
    L.peers_loaded = true;
  }
#endif
  if (t_loaded && RGB_KNOB(dev, 16u)) {
    /* profiling: force the state round trip to complete here */
#ifndef RGB_HOST_EMULATION
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(h0.x), "v"(h6.y) : "memory");
#endif
    *t_loaded = wall_clock64();
  }
  L.ct = h0.x; L.pk = h0.y; L.ci = h1.x; L.la = h1.y; L.li = h2.x; L.lt = h2.y; L.lwi = h3.x;
  L.lwt = h3.y; L.si = h4.x; L.st = h4.y; L.first = h5.x; L.lrs = h5.y; L.lrt = h6.x;
  L.prs = h6.y; L.prt = h7.x; L.pend = h7.y; L.vote_reqs = false;
  /* pre-vote token and machine versions live in the cold qry row: only the election kinds use them */
  constexpr bool TM = KIND < 0 || KIND == RGB_MSG_ELECTION_TIMEOUT || KIND == RGB_MSG_PRE_VOTE_RPC ||
                      KIND == RGB_MSG_PRE_VOTE_RESULT;
  u64 token0 = 0;
  L.token = 0; L.macver = 0;
  if (TM && (L.kind == RGB_MSG_ELECTION_TIMEOUT || L.kind == RGB_MSG_PRE_VOTE_RPC || L.kind == RGB_MSG_PRE_VOTE_RESULT)) {
    const ulonglong2 tm = ldg16(TR, reinterpret_cast<const ulonglong2 *>(qry_row(L) + QRY_TOKEN));
    L.token = token0 = tm.x; L.macver = tm.y;
  }
  L.flags = 0; L.inv = 0; L.has_reply = false; L.reply_to = RGB_NONE;
  L.r_term = L.r_next = L.r_last = L.r_lterm = 0; L.w_first = L.w_last = 0;
  L.n_runs = (unsigned)pk_get(L.pk, PK_NRUNS_SH, 5);
  L.push_cnt = 0;
  L.cond_dirty = false; L.cr0 = L.cr1 = L.cr2 = L.cr3 = 0;
  L.po_floor = 0; L.po_cut = UNDEF;

  const unsigned role0 = role_of(L);
  const unsigned pkhi0 = (unsigned)(L.pk >> 56);      /* the packed word's flag bits 56.. as loaded */
  const u64 ci0 = L.ci, la0 = L.la;
  const unsigned n_runs0 = L.n_runs;
  unsigned n_rpcs = 0;
  int rc = 0;
  /* {next_event, Msg} re-processing of the reference: await_condition hands the message to the condition's
   * transition_to (follower, or leader for the leader's wal_down condition), every other role change that re-processes
   * lands in follower.  So: await_condition first, then the other non-follower roles, then handle_follower once for
   * servers that are (or just became) followers -- one call site per handler */
  unsigned role1 = role0;
  bool go = true;
  if (role0 == RGB_ROLE_AWAIT_CONDITION) {
    bool reprocess = false;
    rc = handle_await_condition<N>(L, reprocess, dev.cond + (size_t)L.server * 4);
    go = !rc && reprocess;
    if (go) { L.flags |= RGB_F_REPROCESSED; role1 = role_of(L); }
  }
  if (go && role1 != RGB_ROLE_FOLLOWER) {
    bool reprocess = false;
    switch (role1) {
      case RGB_ROLE_LEADER:          rc = handle_leader<N, PL>(L, reprocess, dev, rpcs,
                                                           (rpc_slot_base + i) * (N > 1 ? N - 1 : 1),
                                                           msg_index_base + i, n_rpcs); break;
      case RGB_ROLE_CANDIDATE:       rc = handle_candidate<N>(L, reprocess); break;
      case RGB_ROLE_PRE_VOTE:        rc = handle_pre_vote<N>(L, reprocess); break;
      default: L.flags |= RGB_F_UNHANDLED; break;
    }
    go = !rc && reprocess;
    if (go) L.flags |= RGB_F_REPROCESSED;
  }
  if (go) rc = handle_follower<N>(L);
#ifdef RGB_PROFILE
  if (t_loaded && RGB_KNOB(dev, 16u)) t_loaded[2] = wall_clock64();     /* clause code done (stores of rpc records issued) */
#endif
  if (rc) {
    /* the reference would exit/assert: nothing is committed */
    make_decision(out, L.server, role0, RGB_NONE, 0, L.kind, RGB_F_INVARIANT, (u32)rc, 0, 0, 0, 0, ci0, la0);
    return;
  }

  /* ---- commit: run table ---- */
#if RGB_X_COMMIT_WAIT && !defined(RGB_HOST_EMULATION)
  /* every load of the clause code has been consumed by now (the new state depends on them); saying so keeps the
   * compiler from putting a vmcnt(0) -- which on gfx950 also waits for the STORES below to be acknowledged -- in
   * front of the first use of the decision words after the function returns */
  __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
  if (RGB_KNOB(dev, 1u)) { L.n_runs = n_runs0; L.push_cnt = 0; L.cond_dirty = false; L.dmi = L.dni = L.dcs = 0; L.dcs_ci = 0; L.pdirty = 0; }
  if (L.n_runs != n_runs0 || L.push_cnt) {
    u64 *runs = const_cast<u64 *>(L.runs);
    unsigned nr = L.n_runs;
    if (nr > L.max_runs) {
      /* overflow: drop the oldest run(s); the range now starts at the new first run */
      unsigned shift = nr - L.max_runs;
      unsigned in_mem = nr - L.push_cnt;
      for (unsigned k = shift; k < in_mem; ++k) {
        const u64 rs = ldg8(TR, runs + 2 * k), rt = ldg8(TR, runs + 2 * k + 1);
        runs[2 * (k - shift)] = rs;
        runs[2 * (k - shift) + 1] = rt;
      }
      nr = L.max_runs;
      L.flags |= RGB_F_RUNS_OVERFLOW;
      /* new first index = start of the new oldest run */
      u64 nf;
      if (L.push_cnt >= nr) nf = (L.push_cnt == 2 && nr == 2) ? L.prs : L.lrs;
      else nf = ldg8(TR, runs);
      L.first = nf;
      L.n_runs = nr;
    }
    if (L.push_cnt == 2) { runs[2 * (nr - 2)] = L.prs; runs[2 * (nr - 2) + 1] = L.prt; }
    if (L.push_cnt >= 1) { runs[2 * (nr - 1)] = L.lrs; runs[2 * (nr - 1) + 1] = L.lrt; }
  }
  L.pk = pk_set(L.pk, PK_NRUNS_SH, 5, L.n_runs);
  if (L.cond_dirty) {
    ulonglong2 *cp = reinterpret_cast<ulonglong2 *>(dev.cond + (size_t)L.server * 4);
    ST16(cp, make_ulonglong2(L.cr0, L.cr1));
    ST16(cp + 1, make_ulonglong2(L.cr2, L.cr3));
  }
  /* ---- commit: query row (rare) ---- */
  const bool q_reset = ((pkhi0 >> (PK_QPEER_SH - 56)) & 1u) && !pk_get(L.pk, PK_QPEER_SH, 1);
  if ((L.q_dirty || q_reset) && !RGB_KNOB(dev, 1u)) {
    u64 *q = qry_row(L);
    if (L.q_dirty & 1u) q[0] = L.qself;
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (q_reset || (L.q_dirty & (2u << i))) q[1 + i] = L.qp[i];
  }
  /* ---- commit: peers row (dirty words only) ---- */
  if (PL) {
    if (L.pdirty | L.dcs_ci) {
#pragma unroll
      for (int w = 0; w < 2 * N; ++w)
        if (L.pdirty & (1u << w)) ST8(L.peers + w, prow_get(L, (unsigned)w));
#pragma unroll
      for (int k = 0; k < N; ++k)
        if (L.dcs_ci & (1u << k)) ST8(L.peers + PEER_CS(k, N), L.ci);
    }
  } else if (L.dmi | L.dni | L.dcs | L.dcs_ci) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
      if (L.dmi & (1u << k)) ST8(L.peers + PEER_MI(k, N), L.pmi[k]);
      if (L.dni & (1u << k)) ST8(L.peers + PEER_NI(k, N), L.pni[k]);
      if (L.dcs & (1u << k)) ST8(L.peers + PEER_CS(k, N), L.pcs[k]);
      else if (L.dcs_ci & (1u << k)) ST8(L.peers + PEER_CS(k, N), L.ci);
    }
  }
  /* ---- commit: sparse `pending` (rare: only servers uploaded after a write_sparse) ---- */
  if (((pkhi0 >> (PK_PENDX_SH - 56)) & 1u) && !RGB_KNOB(dev, 1u)) {
    u64 *q = qry_row(L);
    u64 s0 = ldg8(TR, q + QRY_PEND_LO), e0 = ldg8(TR, q + QRY_PEND_LO + 1), s1 = ldg8(TR, q + QRY_PEND_HI),
        e1 = ldg8(TR, q + QRY_PEND_HI + 1);
    /* cut: keep [po_floor, po_cut) */
    if (s0 < L.po_floor) s0 = L.po_floor;
    if (s1 < L.po_floor) s1 = L.po_floor;
    if (L.po_cut != UNDEF) {
      if (L.po_cut == 0) { s0 = 1; e0 = 0; s1 = 1; e1 = 0; }
      else { if (e0 >= L.po_cut) e0 = L.po_cut - 1; if (e1 >= L.po_cut) e1 = L.po_cut - 1; }
    }
    bool v0 = s0 <= e0, v1 = s1 <= e1;
    /* canonical form: ranges ascending and non-adjacent, the one that ends at last_index is the newest range */
    const bool newest = range_nonempty(L) && L.pend <= L.li;
    if (v1 && newest && e1 + 1 == L.pend) { L.pend = s1; v1 = false; }          /* ra_seq:append merged them */
    else if (v1 && !newest && range_nonempty(L) && e1 == L.li) { L.pend = s1; v1 = false; }
    else if (!v1 && v0 && newest && e0 + 1 == L.pend) { L.pend = s0; v0 = false; }
    else if (!v1 && v0 && !newest && range_nonempty(L) && e0 == L.li) { L.pend = s0; v0 = false; }
    if (!v1 && v0) { s1 = s0; e1 = e0; v1 = true; v0 = false; }                  /* a single old range sits in HI */
    q[QRY_PEND_LO] = v0 ? s0 : 1; q[QRY_PEND_LO + 1] = v0 ? e0 : 0;
    q[QRY_PEND_HI] = v1 ? s1 : 1; q[QRY_PEND_HI + 1] = v1 ? e1 : 0;
    L.pk = pk_set(L.pk, PK_PENDX_SH, 1, v1 ? 1 : 0);
  }
  /* ---- commit: hot line (only the 16-B pieces that changed) ---- */
  ulonglong2 *ho = reinterpret_cast<ulonglong2 *>(hot);
  if (!RGB_KNOB(dev, 1u)) {
  if (L.n_runs < 2) { L.prs = 0; L.prt = 0; }         /* canonical: no run n-2 */
#ifndef RGB_X_REREAD
#define RGB_X_REREAD 0
#endif
  if ((TR ? RGB_X_REREAD : RGB_X_REREAD_TICK) && PRE) {
    /* what the row held: re-read from the LDS row (still intact) instead of kept in sixteen registers across the
     * clause code */
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const ulonglong2 o = pre[(unsigned)k ^ swz];
      ulonglong2 n;
      switch (k) {
        case HOT_P_TERM: n = make_ulonglong2(L.ct, L.pk); break;
        case HOT_P_CI: n = make_ulonglong2(L.ci, L.la); break;
        case HOT_P_LI: n = make_ulonglong2(L.li, L.lt); break;
        case HOT_P_LW: n = make_ulonglong2(L.lwi, L.lwt); break;
        case HOT_P_SI: n = make_ulonglong2(L.si, L.st); break;
        case HOT_P_FIRST: n = make_ulonglong2(L.first, L.lrs); break;
        case HOT_P_LRT: n = make_ulonglong2(L.lrt, L.prs); break;
        default: n = make_ulonglong2(L.prt, L.pend); break;
      }
      if (n.x != o.x || n.y != o.y) ST16(ho + k, n);
    }
  } else {
  if (L.ct != h0.x || L.pk != h0.y) ST16(ho + HOT_P_TERM, make_ulonglong2(L.ct, L.pk));
  if (L.ci != h1.x || L.la != h1.y) ST16(ho + HOT_P_CI, make_ulonglong2(L.ci, L.la));
  if (L.li != h2.x || L.lt != h2.y) ST16(ho + HOT_P_LI, make_ulonglong2(L.li, L.lt));
  if (L.lwi != h3.x || L.lwt != h3.y) ST16(ho + HOT_P_LW, make_ulonglong2(L.lwi, L.lwt));
  if (L.si != h4.x || L.st != h4.y) ST16(ho + HOT_P_SI, make_ulonglong2(L.si, L.st));
  if (L.first != h5.x || L.lrs != h5.y) ST16(ho + HOT_P_FIRST, make_ulonglong2(L.first, L.lrs));
  if (L.lrt != h6.x || L.prs != h6.y) ST16(ho + HOT_P_LRT, make_ulonglong2(L.lrt, L.prs));
  if (L.prt != h7.x || L.pend != h7.y) ST16(ho + HOT_P_PEND, make_ulonglong2(L.prt, L.pend));
  }
  if (TM && L.token != token0) ST8(qry_row(L) + QRY_TOKEN, L.token);
  }

  u64 w2 = 0, w3 = 0, w4 = 0, w5 = 0;
  if (L.has_reply || L.vote_reqs) { w2 = L.r_term; w3 = L.r_next; w4 = L.r_last; w5 = L.r_lterm; }
  else if (L.flags & RGB_F_WROTE) { w3 = L.w_first; w4 = L.w_last; }
  if ((L.flags & RGB_F_SEND_HEARTBEATS) || L.kind == RGB_MSG_CONSISTENT_QUERY) { w2 = L.hb_term; w5 = L.hb_qi; }
  if (L.flags & RGB_F_QUERY_QUORUM) w3 = L.q_consensus;
  make_decision(out, L.server, role_of(L), L.has_reply ? L.reply_to : (unsigned)RGB_NONE, n_rpcs, L.kind,
                L.flags, 0, w2, w3, w4, w5, L.ci, L.la, L.hb_mask, L.cancel_mask);
#ifdef RGB_PROFILE
  if (t_loaded) { t_loaded[1] = L.prof_nloads; if (RGB_KNOB(dev, 16u)) t_loaded[3] = wall_clock64(); }   /* commit issued */
#endif
}

/* RGB_CFG_FUSE_PIPELINE (opt-in, include/ra_gpu_batch.h).  A leader's same-term success reply and its own written
 * event end with {next_event, info, pipeline_rpcs} (src/ra_server.erl:552, 744), which the gen_statem handles before
 * anything else in its mailbox (:793-801).  Fused, that event runs HERE, right behind the decision that asked for it:
 * literally the RGB_MSG_PIPELINE_RPCS message the host would have submitted next, through the same clause code
 * (process_message<.., RGB_MSG_PIPELINE_RPCS>, rows read back from memory: the first decision has committed), and the
 * two decisions are merged -- n_rpcs and the rpc slots of the event, its flags (RGB_F_PIPELINE again only when the event
 * re-arms itself, RGB_F_SEND_SNAPSHOT, ..) over the reply's.  An event that fails a reference assertion is NOT merged:
 * the reply keeps RGB_F_PIPELINE and the host's message reports the invariant as always.  By construction the state and
 * the records are those of the two steps.  Outside the class paths on purpose: inside handle_leader the extra live
 * ranges cost the per-tick class kernel its register budget (66 spilled VGPRs) whether the mode was on or not. */
template <int N, bool TR>
__device__ __forceinline__ void fuse_pipeline_step(const rgb_dev &dev, Dec &d, u32 i, rgb_rpc *__restrict__ rpcs,
                                                   u32 rpc_slot_base, u32 msg_index_base) {
#ifndef RGB_HOST_EMULATION
  __builtin_amdgcn_s_waitcnt(0x0F70);                     /* the first decision's state stores have been acknowledged */
  asm volatile("" ::: "memory");
#endif
  const u64 m0x = (d.w[0] & 0xFFFFFFFFull) | ((u64)RGB_MSG_PIPELINE_RPCS << 32) | ((u64)RGB_NONE << 40);
  Dec e;
  process_message<N, RGB_MSG_PIPELINE_RPCS, false, TR>(dev, make_ulonglong2(m0x, 0), make_ulonglong2(0, 0), make_ulonglong2(0, 0),
                                                        make_ulonglong2(0, 0), i, rpcs, rpc_slot_base, msg_index_base, e);
  const u32 ef = (u32)e.w[1];
  if (ef & (RGB_F_INVARIANT | RGB_F_UNHANDLED)) return;
  d.w[0] = (d.w[0] & ~(0xFFull << 48)) | (e.w[0] & (0xFFull << 48));                       /* n_rpcs */
  d.w[1] = (d.w[1] & ~(u64)RGB_F_PIPELINE) | (u64)ef;                                      /* flags */
}
__device__ __forceinline__ bool fuse_wanted(const rgb_dev &dev, const Dec &d) {
  const unsigned kind = (unsigned)(d.w[0] >> 56), role = (unsigned)((d.w[0] >> 32) & 0xFF);
  return dev.fuse_pipeline && ((u32)d.w[1] & RGB_F_PIPELINE) && !((u32)d.w[1] & RGB_F_INVARIANT) && role == RGB_ROLE_LEADER &&
         (kind == RGB_MSG_AER_REPLY || kind == RGB_MSG_WRITTEN);
}

/* ------------------------------------------------------------------ fast paths ----
 * The three bulk kinds have ONE steady-state outcome each that takes a few dozen instructions, against the ~1000
 * the general clause code executes per wavefront (every rare clause costs its region's bookkeeping even when no
 * lane enters it).  fast_*() handle exactly that outcome: they test its preconditions on the hot row, and when all
 * hold they produce the decision and the state changes process_message<N, KIND> would have produced, bit for bit
 * (the parity tests run every stream through both); when any fails they touch nothing and return false.
 * Row pieces are read from the lane's LDS row (piece p at pre[p ^ swz]). */
#ifndef RGB_X_FAST
#define RGB_X_FAST 1
#endif
/* -DRGB_X_DECLINE_HIST (tools/decline_hist.py): why a lane declined its fast path, counted per class and reason in
 * dev.dbg_buf (word 0 of every class = lanes that took the fast path) */
#if defined(RGB_X_DECLINE_HIST) && !defined(RGB_HOST_EMULATION)
#define FP_DECLINE(cls, code) do { atomicAdd(dev.dbg_buf + (cls) * 32 + (code), 1ull); return false; } while (0)
#define FP_TAKEN(cls) atomicAdd(dev.dbg_buf + (cls) * 32, 1ull)
#else
#define FP_DECLINE(cls, code) return false
#define FP_TAKEN(cls) do { } while (0)
#endif

/* The steady-state outcomes of append_entries_rpc and written dirty TWO neighbouring 16-byte pieces of the hot row
 * (commit / last_index, last_written / pending: one aligned 32-byte unit each).  Stored by the lane itself they are two
 * instructions that touch 64 lines each; the kernel is bound by the number of its memory requests, so the fast paths
 * only record what they want stored and the wavefront stores it in PAIRS of lanes: in each of two instructions lanes
 * 2k and 2k+1 write the two pieces of ONE server -- 32 contiguous bytes, one request -- so an instruction touches 32
 * lines.  (lanes that took no fast path record nothing) */
#ifndef RGB_X_PAIR_STORE
#define RGB_X_PAIR_STORE 1
#endif
struct PairSt {
  ulonglong2 *p;          /* the first piece of the unit */
  ulonglong2 a, b;
  unsigned m;             /* bit 0: store a at p, bit 1: store b at p + 1 */
};
__device__ __forceinline__ void pair_record(PairSt *ps, ulonglong2 *p, bool sa, const ulonglong2 a, bool sb, const ulonglong2 b) {
  if (RGB_X_PAIR_STORE && ps != nullptr) {
    ps->p = p; ps->a = a; ps->b = b; ps->m = (sa ? 1u : 0u) | (sb ? 2u : 0u);
  } else {
    if (sa) ST16(p, a);
    if (sb) ST16(p + 1, b);
  }
}
__device__ __forceinline__ u64 shfl64(u64 v, int src) {
  return (u64)(u32)__shfl((int)(u32)v, src, 64) | ((u64)(u32)__shfl((int)(u32)(v >> 32), src, 64) << 32);
}
/* executed by the whole wavefront (uniform control flow) */
__device__ __forceinline__ void pair_store(const PairSt &st, u32 lane) {
  const int t = (int)(threadIdx.x & ~63u) + (int)(lane & ~1u);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    /* the unit of lane 2j + k: lane 2j stores its first piece, lane 2j + 1 its second */
    const int src = t + k;
    const u64 pp = shfl64((u64)(uintptr_t)st.p, src);
    const unsigned m = (unsigned)__shfl((int)st.m, src, 64);
    const bool second = (lane & 1u) != 0u;
    const u64 ax = shfl64(st.a.x, src), ay = shfl64(st.a.y, src), bx = shfl64(st.b.x, src), by = shfl64(st.b.y, src);
    if (m & (second ? 2u : 1u))
      ST16(reinterpret_cast<ulonglong2 *>((uintptr_t)pp) + (second ? 1 : 0), second ? make_ulonglong2(bx, by) : make_ulonglong2(ax, ay));
  }
}

/* follower, append_entries_rpc from the known leader in the current term, appended right after the last index with
 * entries of the last run's term: ra_log:write of the tail (src/ra_server.erl:1283-1303, 1365-1389; ra_log:write/2
 * src/ra_log.erl:547-599; evaluate_commit_index_follower/2 :2246-2280) */
__device__ __forceinline__ bool fast_aer(const rgb_dev &dev, const ulonglong2 m0, const ulonglong2 m1, const ulonglong2 m2,
                                         const ulonglong2 m3, const ulonglong2 *pre, unsigned swz, Dec &out,
                                         PairSt *pst = nullptr) {
  const u32 server = (u32)(m0.x & 0xFFFFFFFFull);
  const unsigned wire_kind = (unsigned)((m0.x >> 32) & 0xFF), from = (unsigned)((m0.x >> 40) & 0xFF);
  const unsigned mflags = (unsigned)((m0.x >> 48) & 0xFF), gap = (unsigned)((m0.x >> 56) & 0xFF);
  const u32 n_entries = (u32)(m2.y & 0xFFFFFFFFull), n_run0 = (u32)(m2.y >> 32);
  if (wire_kind != RGB_MSG_AER || server >= dev.n_servers || from >= 8u || mflags != 0) FP_DECLINE(0, 1);
  if (gap != 0) FP_DECLINE(0, 2);
  if (n_run0 < n_entries) FP_DECLINE(0, 4);                          /* entries of two terms */
  const ulonglong2 h0 = pre[HOT_P_TERM ^ swz], h1 = pre[HOT_P_CI ^ swz], h2 = pre[HOT_P_LI ^ swz], h3 = pre[HOT_P_LW ^ swz],
                   h5 = pre[HOT_P_FIRST ^ swz], h6 = pre[HOT_P_LRT ^ swz], h7 = pre[HOT_P_PEND ^ swz];
  const u64 ct = h0.x, pk = h0.y, la = h1.y, li = h2.x, lt = h2.y, lwi = h3.x, first = h5.x, lrs = h5.y, lrt = h6.x,
            pend = h7.y;
  /* follower (role 0, no condition), the sender is the leader we know, nothing sparse pending, a run table */
  if (pk_get(pk, PK_ROLE_SH, 3) != RGB_ROLE_FOLLOWER) FP_DECLINE(0, 5);
  if (pk_get(pk, PK_LEADER_SH, 4) != from) FP_DECLINE(0, 6);
  if (pk_get(pk, PK_PENDX_SH, 1) || pk_get(pk, PK_NRUNS_SH, 5) == 0) FP_DECLINE(0, 7);
  const u64 term = m0.y, pli = m1.x, plt = m1.y, leader_commit = m2.x, eterm = m3.x;
  if (term != ct) FP_DECLINE(0, 8);
  if (!(first <= li) || li == UNDEF) FP_DECLINE(0, 9);
  if (pli != li) FP_DECLINE(0, 10);                                  /* not at the tail: gap, resend, overlap */
  if (li < lrs) FP_DECLINE(0, 11);
  if (plt != lrt) FP_DECLINE(0, 12);                                 /* wrong prev_log_term */
  if (lt != lrt) FP_DECLINE(0, 13);
  if (n_entries != 0 && eterm != lrt) FP_DECLINE(0, 14);             /* entries of a new term: a run is pushed */
  if (la > li + 1 || lwi > li || pend > li + 1) FP_DECLINE(0, 15);
  if (pk_get(pk, PK_NRUNS_SH, 5) < 2 && (h6.y | h7.x) != 0) FP_DECLINE(0, 16);   /* the commit would canonicalise the row */
  FP_TAKEN(0);
  /* has_log_entry_or_snapshot: entry_ok; drop_existing: nothing exists above the last index; ra_log:write extends
   * the last run: last_index moves, last_written and pending keep their words ([pend..] simply grows) */
  const u64 fst = li + 1, lst = li + n_entries;
  u32 flags = RGB_F_LEADER_MSG;
  const u64 ci = leader_commit;
  u64 nla = la;
  const u64 at = lst < ci ? lst : ci;
  if (at > la) { nla = at; flags |= RGB_F_APPLIED | RGB_F_AUX_EVAL; }     /* apply_to/5: at >= la + 1 */
  ulonglong2 *ho = reinterpret_cast<ulonglong2 *>(dev.hot + (size_t)server * RGB_HOT_WORDS);
  static_assert(HOT_P_LI == HOT_P_CI + 1 && (HOT_P_CI & 1) == 0, "commit and last_index pieces: one aligned 32-byte unit");
  pair_record(pst, ho + HOT_P_CI, ci != h1.x || nla != la, make_ulonglong2(ci, nla), n_entries != 0, make_ulonglong2(li + n_entries, lt));
  if (n_entries == 0) {
    /* the empty rpc at the tail (nothing new to write, :1304-1343): validated, the commit index is the leader's, the
     * success reply carries what is written (append_entries_reply/3 :3624-3631) */
    make_decision(out, server, RGB_ROLE_FOLLOWER, from, 0, RGB_MSG_AER, flags | RGB_F_REPLY | RGB_F_REPLY_SUCCESS, 0, term,
                  li + 1, lwi, h3.y, ci, nla);
    return true;
  }
  flags |= RGB_F_WROTE;
  make_decision(out, server, RGB_ROLE_FOLLOWER, RGB_NONE, 0, RGB_MSG_AER, flags, 0, 0, fst, lst, 0, ci, nla);
  return true;
}

/* follower, {written, Term, [From..To]} that confirms (a prefix of) the pending tail inside the last term run:
 * last_written moves to To, the reply goes to the known leader (src/ra_server.erl:1457-1474; ra_log:handle_event
 * src/ra_log.erl:897-920) */
__device__ __forceinline__ bool fast_written(const rgb_dev &dev, const ulonglong2 m0, const ulonglong2 m1,
                                             const ulonglong2 *pre, unsigned swz, Dec &out, PairSt *pst = nullptr) {
  const u32 server = (u32)(m0.x & 0xFFFFFFFFull);
  const unsigned wire_kind = (unsigned)((m0.x >> 32) & 0xFF), mflags = (unsigned)((m0.x >> 48) & 0xFF);
  if (wire_kind != RGB_MSG_WRITTEN || server >= dev.n_servers || mflags != 0) FP_DECLINE(2, 1);
  const ulonglong2 h0 = pre[HOT_P_TERM ^ swz], h1 = pre[HOT_P_CI ^ swz], h2 = pre[HOT_P_LI ^ swz], h3 = pre[HOT_P_LW ^ swz],
                   h4 = pre[HOT_P_SI ^ swz], h5 = pre[HOT_P_FIRST ^ swz], h6 = pre[HOT_P_LRT ^ swz], h7 = pre[HOT_P_PEND ^ swz];
  const u64 ct = h0.x, pk = h0.y, li = h2.x, si = h4.x, first = h5.x, lrs = h5.y, lrt = h6.x, pend = h7.y;
  const unsigned l4 = (unsigned)pk_get(pk, PK_LEADER_SH, 4);
  if (pk_get(pk, PK_ROLE_SH, 3) != RGB_ROLE_FOLLOWER) FP_DECLINE(2, 2);   /* the leader's own written event */
  if (l4 == SLOT_NONE4) FP_DECLINE(2, 3);
  if (pk_get(pk, PK_PENDX_SH, 1) || pk_get(pk, PK_NRUNS_SH, 5) == 0) FP_DECLINE(2, 4);
  const u64 term = m0.y, from = m1.x, to = m1.y;
  /* the whole of [from..to] ends inside the last run, which has the event's term; the snapshot lies below the range */
  if (!(first <= li) || li == UNDEF || from > to) FP_DECLINE(2, 5);
  if (to > li) FP_DECLINE(2, 6);
  if (to < lrs) FP_DECLINE(2, 7);                                     /* ends below the last run */
  if (to < first) FP_DECLINE(2, 8);
  if (lrt != term) FP_DECLINE(2, 9);                                  /* a stale event */
  if (si != UNDEF && si >= first) FP_DECLINE(2, 10);
  if (pend > li + 1) FP_DECLINE(2, 11);
  if (pk_get(pk, PK_NRUNS_SH, 5) < 2 && (h6.y | h7.x) != 0) FP_DECLINE(2, 12);   /* the commit would canonicalise the row */
  /* ra_seq:remove_prefix: the pending tail [pend..li] up to `to` must start inside [from..to] */
  if (pend <= li && pend <= to && pend < from) FP_DECLINE(2, 13);    /* not a prefix: the resend path */
  FP_TAKEN(2);
  u64 npend = pend;
  if (pend <= li && to + 1 > pend) npend = to + 1;                    /* == li + 1 when everything is confirmed */
  const bool changed = !(h3.x == to && h3.y == term);
  ulonglong2 *ho = reinterpret_cast<ulonglong2 *>(dev.hot + (size_t)server * RGB_HOT_WORDS);
  static_assert(HOT_P_PEND == HOT_P_LW + 1 && (HOT_P_LW & 1) == 0, "last_written and pending pieces: one aligned 32-byte unit");
  pair_record(pst, ho + HOT_P_LW, changed, make_ulonglong2(to, term), npend != pend, make_ulonglong2(h7.x, npend));
  u32 flags = 0;
  u64 w2 = 0, w3 = 0, w4 = 0, w5 = 0;
  unsigned reply_to = RGB_NONE;
  if (changed) {                                                      /* append_entries_reply/3 :3624-3631 */
    flags = RGB_F_REPLY | RGB_F_REPLY_SUCCESS;
    w2 = ct; w3 = li + 1; w4 = to; w5 = term; reply_to = slot4to8(l4);
  }
  make_decision(out, server, RGB_ROLE_FOLLOWER, reply_to, 0, RGB_MSG_WRITTEN, flags, 0, w2, w3, w4, w5, h1.x, h1.y);
  return true;
}

/* leader, {Peer, #append_entries_reply{success = true}} of the current term from a member: match / next index of the
 * peer, evaluate_quorum/2, apply (src/ra_server.erl:532-571, 3633-3688); the term of the agreed index must be
 * answerable from the two newest runs (else the general path probes the run table) */
/* PROW: the caller has the peers row in LDS for every lane (prow is not null): the path from memory is not compiled.
 * Round 6: the preconditions are ONE predicate and one branch (eight early returns before: eight exec-mask regions in
 * front of the work), the quorum is a sort (agreed_commit): 762 -> ~450 static instructions, 110 -> ~50 64-bit compares
 * in the train kernel of groups of five -- the reply wavefronts are a quarter of a closed-loop tick's wave-time and
 * four fifths of the literal config 3's, and their clause code is arithmetic. */
template <int N, bool TR = false, bool PROW = false>
__device__ __forceinline__ bool fast_aer_reply(const rgb_dev &dev, const ulonglong2 m0, const ulonglong2 m1,
                                               const ulonglong2 *pre, unsigned swz, Dec &out,
                                               const ulonglong2 *prow = nullptr,
                                               const ulonglong2 *rrow = nullptr) {
  const u32 server = (u32)(m0.x & 0xFFFFFFFFull);
  const unsigned wire_kind = (unsigned)((m0.x >> 32) & 0xFF), peer = (unsigned)((m0.x >> 40) & 0xFF);
  const unsigned mflags = (unsigned)((m0.x >> 48) & 0xFF);
  const ulonglong2 h0 = pre[HOT_P_TERM ^ swz], h1 = pre[HOT_P_CI ^ swz], h2 = pre[HOT_P_LI ^ swz], h3 = pre[HOT_P_LW ^ swz],
                   h4 = pre[HOT_P_SI ^ swz], h5 = pre[HOT_P_FIRST ^ swz], h6 = pre[HOT_P_LRT ^ swz], h7 = pre[HOT_P_PEND ^ swz];
  const u64 ct = h0.x, pk = h0.y, ci0 = h1.x, la = h1.y, li = h2.x, lwi = h3.x, si = h4.x, st = h4.y, first = h5.x,
            lrs = h5.y, lrt = h6.x, prs = h6.y, prt = h7.x;
  const unsigned present = (unsigned)pk_get(pk, PK_PRESENT_SH, 8), voters = (unsigned)pk_get(pk, PK_VOTER_SH, 8);
  const unsigned self = (unsigned)pk_get(pk, PK_SELF_SH, 4), n_runs = (unsigned)pk_get(pk, PK_NRUNS_SH, 5);
#if defined(RGB_X_DECLINE_HIST) && !defined(RGB_HOST_EMULATION)
  if (wire_kind != RGB_MSG_AER_REPLY || server >= dev.n_servers || peer >= (unsigned)N) FP_DECLINE(1, 1);
  if (mflags != RGB_MF_SUCCESS) FP_DECLINE(1, 2);                    /* a failed reply */
  if (dev.fuse_pipeline) FP_DECLINE(1, 9);                           /* opt-in fused pipelining: the general path emits the rpcs */
  if (pk_get(pk, PK_ROLE_SH, 3) != RGB_ROLE_LEADER) FP_DECLINE(1, 3);
  if (m0.y != ct) FP_DECLINE(1, 4);
  if (!((present >> peer) & 1u)) FP_DECLINE(1, 5);
  if (n_runs < 2 && (h6.y | h7.x) != 0) FP_DECLINE(1, 6);
#else
  /* the message is a success reply (not fused: the general path emits the rpcs then) from a member, to a leader, in
   * its current term; a row the commit would not have to canonicalise */
  const bool ok = (wire_kind == RGB_MSG_AER_REPLY) & (server < dev.n_servers) & (peer < (unsigned)N) &
                  (mflags == RGB_MF_SUCCESS) & (dev.fuse_pipeline == 0u) &
                  (pk_get(pk, PK_ROLE_SH, 3) == RGB_ROLE_LEADER) & (m0.y == ct) & (((present >> (peer & 7u)) & 1u) != 0u) &
                  !((n_runs < 2) & ((h6.y | h7.x) != 0));
  if (!ok) return false;
#endif
  u64 *peers = dev.peers + (size_t)server * dev.peer_stride;
  /* piece i of the peers row = (match_index, next_index) of member i */
  u64 wm[N], wn[N];
  if (PROW || prow != nullptr) {
    const unsigned psw = rgb_wide_peers((unsigned)N) ? swz << 1 : swz;      /* (wide rows: 256 bytes of LDS each) */
#pragma unroll
    for (int k = 0; k < N; ++k) { const ulonglong2 v = prow[(unsigned)k ^ psw]; wm[k] = v.x; wn[k] = v.y; }
  } else {
    const ulonglong2 *pp = reinterpret_cast<const ulonglong2 *>(peers);
#pragma unroll
    for (int k = 0; k < N; ++k) { const ulonglong2 v = ldg16(TR, pp + k); wm[k] = v.x; wn[k] = v.y; }
  }
  /* match_index / next_index of the peer only move forward (:540-547) */
  u64 mi_new = 0, ni_new = 0; bool mi_dirty = false, ni_dirty = false;
  if (PROW) {
    /* the peer's own piece once more, by its (per-lane) position: two 64-bit compares instead of N predicated pairs */
    const unsigned psw = rgb_wide_peers((unsigned)N) ? swz << 1 : swz;
    const ulonglong2 pv = prow[peer ^ psw];
    mi_dirty = m1.y > pv.x; ni_dirty = m1.x > pv.y;
    mi_new = mi_dirty ? m1.y : pv.x; ni_new = ni_dirty ? m1.x : pv.y;
#pragma unroll
    for (int i = 0; i < N; ++i) wm[i] = ((unsigned)i == peer) ? mi_new : wm[i];
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if ((unsigned)i != peer) continue;
      mi_new = wm[i]; ni_new = wn[i];
      if (m1.y > wm[i]) { mi_new = m1.y; mi_dirty = true; wm[i] = m1.y; }
      if (m1.x > wn[i]) { ni_new = m1.x; ni_dirty = true; }
    }
  }
  /* agreed_commit/1 over the voters' match indexes and the leader's last written index: descending order statistic
   * n/2 + 1 */
  u64 v[N + 1]; bool use[N + 1]; int n = 1;
  v[N] = lwi; use[N] = true;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const bool u = ((unsigned)i != self) && ((present >> i) & 1u) && ((voters >> i) & 1u);
    v[i] = wm[i]; use[i] = u; n += u ? 1 : 0;
  }
  const u64 p = agreed_commit<N + 1>(v, use, n);
  u64 t = UNDEF;
  if (first <= li && p >= first && p <= li) {
    if (p >= lrs) t = lrt;
    else if (n_runs >= 2 && p >= prs) t = prt;
    else {
      /* older than the mirrored runs: the newest in-memory run (n_runs-3 .. 0) that starts at or below p holds it
       * (ra_log:fetch_term/2 inside the range, src/ra_log.erl:1186-1200).  Train launches: runs 0..7 are in the table
       * line the wavefront fetched into LDS with the rows (run k at rrow[k ^ swz]); the runs behind the line -- a table
       * of more than ten runs: 3 % of the lanes of the closed loop, but two wavefronts in three had one and ran the
       * whole general path for it (round 5) -- and every run of a per-tick launch are read from the table itself,
       * newest first */
      bool found = false;
      const ulonglong2 *rt = reinterpret_cast<const ulonglong2 *>(dev.runs + (size_t)server * dev.max_runs * 2u);
      const int lds_runs = rrow != nullptr ? RGB_RUNS_LDS : 0;
#pragma unroll 1
      for (int k = (int)n_runs - 3; k >= lds_runs && !found; --k) {
        const ulonglong2 r = ldg16(TR, rt + k);
        if (p >= r.x) { t = r.y; found = true; }
      }
      if (rrow != nullptr) {
#pragma unroll
        for (int k = RGB_RUNS_LDS - 1; k >= 0; --k) {
          if (!found && (unsigned)k + 3u <= n_runs) {
            const ulonglong2 r = rrow[(unsigned)k ^ swz];
            if (p >= r.x) { t = r.y; found = true; }
          }
        }
      }
      if (!found) FP_DECLINE(1, 8);
    }
  }
  FP_TAKEN(1);
  if (t == UNDEF && si != UNDEF && si == p) t = st;
  u64 ci = ci0, nla = la;
  u32 flags = RGB_F_PIPELINE;
  if (t != UNDEF && t == ct) ci = p;                                 /* Raft 5.4.2; NO max() */
  if (ci > ci0) flags |= RGB_F_AUX_EVAL;
  if (ci > la) { const u64 to = li < ci ? li : ci; if (to >= la + 1) { nla = to; flags |= RGB_F_APPLIED; } }
  /* the two words are one 16-byte piece of the row: stored whole when either moved (the other keeps its value; a 16-byte
   * and an 8-byte store cost the fabric the same 32-byte write) */
  if (mi_dirty | ni_dirty) ST16(reinterpret_cast<ulonglong2 *>(peers) + peer, make_ulonglong2(mi_new, ni_new));
  if (ci != ci0 || nla != la)
    ST16(reinterpret_cast<ulonglong2 *>(dev.hot + (size_t)server * RGB_HOT_WORDS) + HOT_P_CI, make_ulonglong2(ci, nla));
  make_decision(out, server, RGB_ROLE_LEADER, RGB_NONE, 0, RGB_MSG_AER_REPLY, flags, 0, 0, 0, 0, 0, ci, nla);
  return true;
}

/* The tick kernel: one lane per message, one wavefront per 64 consecutive messages.  Messages
 * come in and decisions go out through LDS so that every global access of the 64-byte records
 * is a fully coalesced 1 KiB wave transaction (lane stride 16 B) instead of 64 strided 16-byte
 * pieces; LDS slots are padded to 80 B so the per-lane 16-byte reads/writes are conflict-free. */
#define RGB_IO_SLOT 5   /* 16-byte units per LDS record slot: 64 B payload + 16 B pad */
#define RGB_HOT_SLOT 9  /* 16-byte units reserved per lane in the class kernel's LDS area (rows use 8, unpadded) */

template <int N, int KIND, bool SEQXP = (KIND < 0)>
__global__ __launch_bounds__(RGB_TICK_BLOCK, RGB_MIN_WAVES(N)) void rgb_tick_kernel(rgb_dev dev, const rgb_msg *__restrict__ msgs,
                                                                  u32 n, const u32 *__restrict__ n_dev,
                                                                  rgb_decision *__restrict__ dec,
                                                                  rgb_rpc *__restrict__ rpcs, u32 rpc_slot_base,
                                                                  u32 msg_index_base) {
  __shared__ ulonglong2 io[RGB_TICK_BLOCK * RGB_IO_SLOT];
  const u32 lane = threadIdx.x;
  const u32 base = blockIdx.x * RGB_TICK_BLOCK;           /* first message of this wavefront */
  if (n_dev != nullptr) {                                 /* the tick's real size lives on the device */
    const u32 nd = *n_dev;
    n = nd < n ? nd : n;
    if (base >= n) return;                                /* uniform per block */
  }
  const u32 cnt = n - base < RGB_TICK_BLOCK ? n - base : RGB_TICK_BLOCK;
  const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(msgs + base);
  {
    const u32 last = cnt * 4u - 1u;                       /* four wave loads in flight; see the class kernel */
    ulonglong2 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32 piece = k * RGB_TICK_BLOCK + lane;        /* 16-byte piece of the 64-message block */
      v[k] = ld16<true>(src + (piece < last ? piece : last));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32 piece = k * RGB_TICK_BLOCK + lane;
      io[(piece >> 2) * RGB_IO_SLOT + (piece & 3u)] = v[k];
    }
  }
  lds_barrier();
  const bool active = lane < cnt;
  Dec d;
  if (active) {
    const ulonglong2 m0 = io[lane * RGB_IO_SLOT + 0], m1 = io[lane * RGB_IO_SLOT + 1],
                     m2 = io[lane * RGB_IO_SLOT + 2], m3 = io[lane * RGB_IO_SLOT + 3];
    process_message<N, KIND, false, false, SEQXP>(dev, m0, m1, m2, m3, base + lane, rpcs, rpc_slot_base, msg_index_base, d);
    if (fuse_wanted(dev, d)) fuse_pipeline_step<N, false>(dev, d, base + lane, rpcs, rpc_slot_base, msg_index_base);
    (void)compact_decision(d);
    io[lane * RGB_IO_SLOT + 0] = make_ulonglong2(d.w[0], d.w[1]);
    io[lane * RGB_IO_SLOT + 1] = make_ulonglong2(d.w[2], d.w[3]);
    io[lane * RGB_IO_SLOT + 2] = make_ulonglong2(d.w[4], d.w[5]);
    io[lane * RGB_IO_SLOT + 3] = make_ulonglong2(d.w[6], d.w[7]);
  }
  lds_barrier();
  if (RGB_KNOB(dev, 2u)) return;
  ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(dec + base);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const u32 piece = k * RGB_TICK_BLOCK + lane;
    const u32 j = piece >> 2, part = piece & 3u;
    if (j < cnt && (part < 2u || !((u32)io[j * RGB_IO_SLOT].y & RGB_F_COMPACT)))
      store16_nt((void *)(dst + piece), io[j * RGB_IO_SLOT + part]);
  }
}


/* The per-tick kernel for family-ordered ticks: ONE launch, each wavefront (= block) serves one
 * 64-message slice of one kernel class and runs the code path specialised for that class's
 * message kind (compile-time kind => the clause switches fold; every path needs <= 128 VGPRs, so
 * four wavefronts per SIMD stay resident, against two for the kind-generic kernel).  One class per
 * message kind, in family order.  Blocks are handed to classes heaviest-first (pipelining kinds,
 * elections, replies, then the cheap append_entries_rpc / written bulk) so the long paths start
 * on an idle memory system and finish under the cover of the bulk.
 *
 * The block -> (class, slice) map is a PLAN: 15 cumulative block counts in heaviest-first order plus the
 * class offsets and sizes.  The host builds it when it knows the class sizes (it travels in the kernel
 * arguments: the lookup is 14 scalar compares and three scalar loads); for device-produced ticks the
 * kernel derives the same plan from the per-family totals in device memory. */
struct rgb_tick_plan {
  u32 blk_end[RGB_N_CLASSES];   /* blocks of positions 0..q, cumulative            */
  u32 off[RGB_N_CLASSES];       /* first message of the class at position q         */
  u32 cnt[RGB_N_CLASSES];       /* messages of the class at position q              */
};

/* heaviest classes first (ranks: 3 append, 4 pipeline_rpcs, 9 pre_vote_rpc, 8 election_timeout,
 * 10 pre_vote_result, 6 vote_result, 5 request_vote, 7 await_timeout, 11 snapshot_written,
 * 14 consistent_query, 13 heartbeat_reply, 12 heartbeat_rpc, 1 append_entries_reply,
 * 0 append_entries_rpc, 2 written), 4 bits per position */
#define RGB_CLASS_ORDER 0x201CDEB756A8943ull
__host__ __device__ constexpr int rgb_class_at(unsigned q) { return (int)((RGB_CLASS_ORDER >> (4u * q)) & 0xFu); }
static_assert(rgb_class_at(0) == 3 && rgb_class_at(1) == 4 && rgb_class_at(2) == 9 && rgb_class_at(3) == 8 &&
              rgb_class_at(4) == 10 && rgb_class_at(5) == 6 && rgb_class_at(6) == 5 && rgb_class_at(7) == 7 &&
              rgb_class_at(8) == 11 && rgb_class_at(9) == 14 && rgb_class_at(10) == 13 && rgb_class_at(11) == 12 &&
              rgb_class_at(12) == 1 && rgb_class_at(13) == 0 && rgb_class_at(14) == 2, "class order");

/* n[c] = messages of class c (family order in memory) */
__host__ __device__ __forceinline__ void rgb_make_plan(const u32 (&n)[RGB_N_CLASSES], rgb_tick_plan &p, unsigned n_members) {
  u32 blocks = 0;
#pragma unroll
  for (int q = 0; q < RGB_N_CLASSES; ++q) {
    const int c = rgb_class_at((unsigned)q);
    u32 off = 0, cnt = 0;
#pragma unroll
    for (int k = 0; k < RGB_N_CLASSES; ++k) { off += k < c ? n[k] : 0u; cnt = k == c ? n[k] : cnt; }
    const u32 sl = rgb_class_slice(c, n_members);
    blocks += (cnt + sl - 1) / sl;
    p.blk_end[q] = blocks; p.off[q] = off; p.cnt[q] = cnt;
  }
}

__device__ __forceinline__ u32 rgb_xcc_id() {
#ifdef RGB_HOST_EMULATION
  return 0;
#else
  u32 xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
  return xcc;
#endif
}

/* Train launches (rgb_train_kernel): a wavefront whose servers' previous messages have not committed yet polls their
 * sequence stamps this many times (one L2-served load + s_sleep per try, ~1 us) before it gives up and raises
 * RGB_TRAIN_ERR_SPIN -- a bound, so that a broken dependency can never hang the device */
#ifndef RGB_TRAIN_SPIN_LIMIT
#ifdef RGB_HOST_EMULATION
#define RGB_TRAIN_SPIN_LIMIT 16u      /* blocks run one after another on the CPU: a dependency is met or never will be */
#else
#define RGB_TRAIN_SPIN_LIMIT 20000u    /* x ~4 us per try once backed off: ~80 ms */
#endif
#endif

#ifndef RGB_X_TICKET_AT
#define RGB_X_TICKET_AT 0   /* where a persistent wavefront requests its next row: 0 behind its publish, 1 in front of its
                               clause code (the rows have arrived), 2 at the start of its slice */
#endif
/* Persistent train launches: a wavefront takes the next row of its shard from the shard's ticket counter: one
 * returning atomic by lane 0.  rgb_take_ticket only ISSUES it (the raw value is valid in lane 0); rgb_ticket_value
 * brings it to the whole wavefront where it is consumed, so the atomic's round trip overlaps whatever is issued in
 * between.  The counters of different shards are 128 bytes apart and each is only ever touched from one XCD. */
__device__ __forceinline__ u32 rgb_take_ticket(u32 *ctr, const u32 *err) {
  u32 v = 0;
#ifdef RGB_HOST_EMULATION
  if (threadIdx.x == 0) v = atomicAdd(ctr, 1u);
  if (threadIdx.x == 1) v = *err;
#else
#ifdef RGB_X_TICKET_WG   /* EXPERIMENT: no sc1 on the atomic (all users of a counter share one XCD's L2) */
  if (threadIdx.x == 0) v = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
  if (threadIdx.x == 0) v = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
  /* lane 1 of the same register: the launch's error word (a failed launch drains instead of computing on) */
  if (threadIdx.x == 1) v = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
  return v;
}
__device__ __forceinline__ u32 rgb_ticket_value(u32 raw) {
#ifdef RGB_HOST_EMULATION
  return __shfl(raw, 0, 64);
#else
  return (u32)__builtin_amdgcn_readfirstlane((int)raw);
#endif
}
__device__ __forceinline__ u32 rgb_ticket_err(u32 raw) {
#ifdef RGB_HOST_EMULATION
  return __shfl(raw, 1, 64);
#else
  return (u32)__builtin_amdgcn_readlane((int)raw, 1);
#endif
}

/* One wavefront's slice of one class: `cnt` (<= SL) consecutive messages starting at msgs[base].  This is the whole
 * hot path -- record staging, cooperative state fetch, fast paths, clause code, commit, decision store -- shared by
 * the per-tick class kernel (TR = false) and the multi-tick train kernel (TR = true).
 *
 * TR = true adds the dataflow protocol that replaces the kernel boundary between ticks.  Every server has a sequence
 * byte in dev.seq (the number of messages applied to it, mod 256; shard-major, so an XCD only ever touches its own
 * lines of the array, which stay in its L2) and every message of a train carries the value it must find
 * (stamps[], written by rgb_train_seq_kernel).  The wavefront
 *   1. polls the sequence bytes of its servers until every one matches (the earlier message of each server has
 *      committed -- by a wavefront of an earlier tick of the SAME launch, on the same XCD: see rgb_train_kernel),
 *   2. fetches the rows with L2-served loads (sc1: a CU's L1 is never refreshed by another CU's stores),
 *   3. runs the unchanged clause code, whose state stores are plain (they stay in the XCD's L2),
 *   4. waits for every store to be acknowledged, then stores the advanced sequence byte: whoever sees the new
 *      value sees the whole commit.
 * A launch carries at most 255 ticks, so the values a server goes through within one launch are distinct. */
template <int N, bool TR>
__device__ __forceinline__ bool rgb_tick_slice(const rgb_dev &dev, ulonglong2 *io, const int cls, const u32 base,
                                               const u32 cnt, const u32 SL, const rgb_msg *__restrict__ msgs,
                                               rgb_decision *__restrict__ dec, rgb_rpc *__restrict__ rpcs,
                                               u32 rpc_slot_base, u32 msg_index_base, u32 *__restrict__ ctl,
                                               const unsigned char *__restrict__ stamps,
                                               u32 *__restrict__ ticket_ctr = nullptr, u32 *next_ticket = nullptr,
                                               u32 shard = 0, u32 lane_in = 0, const u32 *place_word = nullptr,
                                               u32 place_bit = 0) {
  const u32 lane = TR ? lane_in : (u32)threadIdx.x;     /* the persistent loop hands in an opaque copy */
#ifdef RGB_PROFILE
  u64 t0 = 0, t1 = 0, t2 = 0, t2b = 0, tl[4] = {0, 0, 0, 0};
  if (!TR && RGB_KNOB(dev, 16u)) t0 = wall_clock64();
#endif
  const bool lead_cls = rgb_lead_class(cls);              /* append_entries_reply, append, pipeline_rpcs */
  constexpr bool PEERS_LDS = rgb_class_slice(1, (unsigned)N) == 32u;
  /* train launches of groups of six to eight members: the leader-side slices are 32 messages as well, their 192-byte
   * peers rows come into LDS cooperatively (rgb_train_class_slice) */
  constexpr bool PEERS_WIDE = TR && rgb_train_class_slice(1, (unsigned)N) == 32u && !PEERS_LDS;
  /* a table row of max_runs >= 8 runs holds the whole 128-byte line that is fetched (wave-uniform) */
  const bool RUNS_LDS = TR && RGB_TRAIN_RUNS_LDS && (PEERS_LDS || PEERS_WIDE) && dev.max_runs >= (u32)RGB_RUNS_LDS;
  /* groups of six and more members (no peers rows in LDS: the area is free): an append_entries_rpc wavefront whose
   * messages point below the current term's entries (prev_log_term != term in any lane: log-matching repair, the
   * configs[4] workload) will walk its servers' run tables -- their first line comes with the hot rows (decided
   * from the messages alone, before any state arrives) and has_log_entry_or_snapshot / drop_existing are served from
   * LDS instead of one dependent L2 round trip per run */
  bool AER_RUNS = false;
#if defined(RGB_X_TRAIN_TIMELINE) && !defined(RGB_HOST_EMULATION)
  /* EXPERIMENT build (tools/train_timeline.py): per-wavefront wall-clock stamps of a train launch */
  u64 tt[7] = {0, 0, 0, 0, 0, 0, 0};
  unsigned tt_spins = 0;
#define RGB_TT(k) do { if (TR) tt[k] = wall_clock64(); } while (0)
#else
#define RGB_TT(k) do { } while (0)
#endif
  RGB_TT(0);
  if (TR && RGB_X_TICKET_AT == 2 && ticket_ctr != nullptr) *next_ticket = rgb_take_ticket(ticket_ctr, ctl);
  const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(msgs + base);
  /* a train's stamp byte travels with the record copies: requested here, it arrives under their round trip (behind
   * the records' barrier it was a second memory round trip in front of the first poll) */
  unsigned need_raw = 0;
  if (TR && lane < cnt) need_raw = (unsigned)stamps[base + lane];
  {
    /* four 1 KiB global -> LDS copies in flight (read once: non-temporal), no staging registers and no ds_write
     * pass: record r lands at io[4 r ..], its piece p at position p ^ ((r >> 2) & 3).  The permutation is applied to
     * the SOURCE address (the instruction's destination is lane-linear) and again when the owner reads its record:
     * conflict-free 16-byte LDS reads without padding.  Pieces past the slice's end re-read its last piece, their
     * LDS slots are never consumed. */
    const u32 last = cnt * 4u - 1u;
    const bool half = rgb_half_msg_class(cls);            /* only pieces 0, 1 of every record are asked for */
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if ((u32)k * 16u >= SL) break;                      /* a 32-message slice is two copies */
      const u32 piece = k * RGB_TICK_BLOCK + lane;
      const u32 r = piece >> 2;
      const u32 sp = (r << 2) | ((piece & 3u) ^ ((r >> 2) & 3u));
      if (!half || (sp & 2u) == 0u) glds16<GLDS_NT>(src + (sp < last ? sp : last), io + k * RGB_TICK_BLOCK);
    }
    glds_wait();
  }
  lds_barrier();
#ifdef RGB_PROFILE
  if (!TR && RGB_KNOB(dev, 16u)) t1 = wall_clock64();
#endif
  const bool active = lane < cnt;
  const u32 mswz = (lane >> 2) & 3u;
  const ulonglong2 m0 = io[lane * 4 + (0 ^ mswz)], m1 = io[lane * 4 + (1 ^ mswz)];
  ulonglong2 m2 = io[lane * 4 + (2 ^ mswz)], m3 = io[lane * 4 + (3 ^ mswz)];
  if (rgb_half_msg_class(cls)) {
    /* the upper half was not loaded: zeros, as the field list of these kinds says -- but for a written event that
     * carries two ranges (RGB_MF_SEQ2: the lower one rides in the record's last piece), which re-reads its record */
    m2 = make_ulonglong2(0, 0); m3 = make_ulonglong2(0, 0);
    if (cls == 2 && active && (((m0.x >> 48) & 0xFFull) & RGB_MF_SEQ2)) m3 = ld16<true>(src + lane * 4u + 3u);
  }
  RGB_TT(1);
  if (TR && RGB_TRAIN_RUNS_LDS && !PEERS_LDS && cls == 0 && dev.max_runs >= (u32)RGB_RUNS_LDS)
    AER_RUNS = __ballot(active && m1.y != m0.y) != 0ull;
  const u32 sv = (u32)(m0.x & 0xFFFFFFFFull);
  /* the message addresses a server (a NOP or an out-of-range id touches no state and has no stamp) */
  const bool has_srv = TR && active && sv < dev.n_servers && ((m0.x >> 32) & 0xFFull) != RGB_MSG_NOP;
  unsigned char *seqp = nullptr;
  unsigned need = 0;
  if (TR) {
    /* 0. every message of the slice belongs to the shard (= the XCD) this wavefront serves: a tick that is not in
     * bucket order must not be computed on another XCD's lines */
    if (__ballot(has_srv && rgb_shard_of_server(sv, (unsigned)N) != shard) != 0ull) {
      if (lane == 0) atomicOr(ctl, (u32)RGB_TRAIN_ERR_ORDER);
      return false;
    }
    /* 1. dependencies: this server's previous message -- an earlier tick of this launch -- has committed */
    seqp = dev.seq + rgb_seq_index(has_srv ? sv : 0u, dev.n_members, dev.seq_stride);
    need = has_srv ? need_raw : 0u;
    unsigned spins = 0;
    bool late = has_srv;
#ifdef RGB_X_TRAIN_NODEPS
    /* EXPERIMENT builds (never in the product; break parity; bench.py --snapshot-kernel: the rows of an in-launch
     * snapshot wait for exact bytes): 1 = no poll at all -- the tick without the dependency waits AND without the
     * poll's round trip (what a sequence tag inside the row could give at best); 2 = one poll whose answer is
     * ignored -- without the waits only.  Round 5, same box: 17.46 us per tick, 16.72 with 2, 16.36 with 1 */
    if (RGB_X_TRAIN_NODEPS == 1 || RGB_X_TRAIN_NODEPS == 3) late = false;
#endif
    for (;;) {
      /* only the lanes that are still waiting poll again; a wavefront that has to wait backs off (thousands of
       * waiting wavefronts polling flat out starve the ones they wait for of L2 bandwidth) */
      if (late) {
#ifdef RGB_HOST_EMULATION
        const unsigned cur = *seqp;
#elif defined(RGB_X_POLL32)
        /* EXPERIMENT (round 5, neutral: 15.95 / 16.13 against 15.94 / 16.02 us per tick): the poll as a load of the byte's
         * DWORD (neighbouring lanes' bytes share dwords and lines) */
        const uintptr_t pa = reinterpret_cast<uintptr_t>(seqp);
        const unsigned cur = (__hip_atomic_load(reinterpret_cast<const u32 *>(pa & ~(uintptr_t)3), __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT) >> (8u * (unsigned)(pa & 3u))) & 0xFFu;
#else
        const unsigned cur = __hip_atomic_load(seqp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        late = cur != need;
#ifdef RGB_X_TRAIN_NODEPS
        if (RGB_X_TRAIN_NODEPS == 2) late = false;
#endif
      }
      if (__ballot(late) == 0ull) break;
      spins += 1;
#if defined(RGB_X_TRAIN_TIMELINE) && !defined(RGB_HOST_EMULATION)
      tt_spins = spins;
#endif
      bool give_up = spins > RGB_TRAIN_SPIN_LIMIT;
#ifndef RGB_HOST_EMULATION
      if ((spins & 15u) == 0u && __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) give_up = true;
#ifndef RGB_TRAIN_SLEEP
#define RGB_TRAIN_SLEEP 4                                 /* x 64 clocks: ~0.1 us between polls.  The closed loop does not care
                                                             (2..32 within noise, rounds 3 and 6); the chain-bound literal config 3,
                                                             whose wavefronts all wait, runs 3 % faster with 4 than with 8 or 16
                                                             (7.10-7.17 against 7.35-7.47 us per tick, same box, round 6) */
#endif
      __builtin_amdgcn_s_sleep(RGB_TRAIN_SLEEP);
      if (spins > 64u) __builtin_amdgcn_s_sleep(127);     /* a long wait (> ~15 us) is not the steady state: back off */
#endif
      if (give_up) {                                      /* uniform: the decisions of this slice stay unwritten */
        if (lane == 0) atomicOr(ctl, (u32)RGB_TRAIN_ERR_SPIN);
        return false;
      }
    }
  }
  /* Cooperative hot-line fetch: 8 lanes read one server's 128-byte line as ONE coalesced access, so
   * an instruction touches 8 lines instead of 64 (the CU's L1 looks up one line per cycle); the lines
   * reach their owners through LDS rows that overlay the record staging area (the messages are in
   * registers by now), and process_message reads its row from LDS piece by piece, when it needs it. */
  RGB_TT(2);
  constexpr bool PRE = true;
  constexpr int ROWS = TR ? GLDS_SC1 : GLDS_DEFAULT;      /* a train's rows are L2-served (see above) */
  u64 pf0 = 0, pf1 = 0;
  lds_barrier();
  {
    const u32 srv = (active && sv < dev.n_servers) ? sv : 0u;
    /* leader-side classes: one word of the peers row is requested in the same round trip as the hot
     * lines, so the row's line(s) are in the cache when the lane loads the row into registers (a second
     * full-latency round trip otherwise; keeping the whole row in registers across the fetch costs
     * spills on the pipelining paths) */
    if (lead_cls && !PEERS_LDS && !PEERS_WIDE) {
      const u64 *pp = dev.peers + (size_t)srv * dev.peer_stride;
      pf0 = ldg8(TR, pp);
      if (3 * N > 16) pf1 = ldg8(TR, pp + 16);
    }
    /* row r = 8k + lane/8 lands at io[8 r ..] (1 KiB per instruction, lane-linear destination); position q of the
     * row holds piece q ^ ((r >> 1) & 7): the 16 lanes the LDS serves together read 16 different bank groups */
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if ((u32)k * 8u >= SL) break;
      const u32 r = 8 * k + (lane >> 3);
      const u32 sj = __shfl(srv, (int)r, 64);
      glds16<ROWS>(reinterpret_cast<const ulonglong2 *>(dev.hot + (size_t)sj * RGB_HOT_WORDS) + ((lane & 7u) ^ ((r >> 1) & 7u)),
                   io + k * RGB_TICK_BLOCK);
    }
    if (!PEERS_LDS && AER_RUNS) {
      /* the first line of the 64 run tables behind the 64 hot rows, same shape */
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const u32 r = 8 * k + (lane >> 3);
        const u32 sj = __shfl(srv, (int)r, 64);
        glds16<ROWS>(reinterpret_cast<const ulonglong2 *>(dev.runs + (size_t)sj * dev.max_runs * 2u) + ((lane & 7u) ^ ((r >> 1) & 7u)),
                     io + (8 + k) * RGB_TICK_BLOCK);
      }
    }
    if (PEERS_WIDE && lead_cls) {
      /* the 192-byte peers rows of the slice's 32 servers: four rows per instruction, 16 lanes a row of which twelve
       * fetch (256 bytes of LDS per row, behind the 32 hot rows); piece p of row r sits at position p ^ 2 ((r >> 1) & 7)
       * -- the owners' 16-byte reads collide in pairs only -- and the first line of the 32 run tables behind them */
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const u32 r = 4 * k + (lane >> 4);
        const u32 sj = __shfl(srv, (int)r, 64);
        const u32 pcx = (lane & 15u) ^ (((r >> 1) & 7u) << 1);
        if (pcx < 12u)
          glds16<ROWS>(reinterpret_cast<const ulonglong2 *>(dev.peers + (size_t)sj * 24u) + pcx, io + (4 + k) * RGB_TICK_BLOCK);
      }
      if (RUNS_LDS) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const u32 r = 8 * k + (lane >> 3);
          const u32 sj = __shfl(srv, (int)r, 64);
          glds16<ROWS>(reinterpret_cast<const ulonglong2 *>(dev.runs + (size_t)sj * dev.max_runs * 2u) + ((lane & 7u) ^ ((r >> 1) & 7u)),
                       io + (12 + k) * RGB_TICK_BLOCK);
        }
      }
    }
    if (PEERS_LDS && lead_cls) {
      /* the peers rows (one 128-byte line each) of the slice's 32 servers behind the 32 hot rows, same shape */
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const u32 r = 8 * k + (lane >> 3);
        const u32 sj = __shfl(srv, (int)r, 64);
        glds16<ROWS>(reinterpret_cast<const ulonglong2 *>(dev.peers + (size_t)sj * 16u) + ((lane & 7u) ^ ((r >> 1) & 7u)),
                     io + (4 + k) * RGB_TICK_BLOCK);
      }
      if (RUNS_LDS) {
        /* train launches: the first line of the 32 run tables behind the peers rows (see run_pair) */
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const u32 r = 8 * k + (lane >> 3);
          const u32 sj = __shfl(srv, (int)r, 64);
          glds16<ROWS>(reinterpret_cast<const ulonglong2 *>(dev.runs + (size_t)sj * dev.max_runs * 2u) + ((lane & 7u) ^ ((r >> 1) & 7u)),
                       io + (8 + k) * RGB_TICK_BLOCK);
        }
      }
    }
    glds_wait();
  }
  lds_barrier();
  RGB_TT(3);
  if (TR && RGB_X_TICKET_AT == 1 && ticket_ctr != nullptr) *next_ticket = rgb_take_ticket(ticket_ctr, ctl);
  const ulonglong2 *hrow = io + lane * 8;
  const unsigned hswz = (lane >> 1) & 7u;
  const ulonglong2 *prow = (PEERS_LDS && lead_cls) ? io + 4 * RGB_TICK_BLOCK + lane * 8
                           : (PEERS_WIDE && lead_cls) ? io + 4 * RGB_TICK_BLOCK + lane * 16 : nullptr;
  const ulonglong2 *rrow = (RUNS_LDS && PEERS_WIDE && lead_cls) ? io + 12 * RGB_TICK_BLOCK + lane * 8
                           : ((RUNS_LDS && PEERS_LDS && lead_cls) || (!PEERS_LDS && AER_RUNS)) ? io + 8 * RGB_TICK_BLOCK + lane * 8 : nullptr;
#ifndef RGB_HOST_EMULATION
  asm volatile("" ::"v"(pf0), "v"(pf1));   /* the touch loads above stay in the program */
#endif
  Dec d;
  u64 *tlp = nullptr;
#ifdef RGB_PROFILE
  if (!TR) tlp = tl;
#endif
  /* deferred rpc records (emit_rpc): a train's leader-side wavefronts (32-message slices: the hot rows lie at the
   * front of the staging area whatever the group size) park the first four records of a message in the lane's own hot
   * row -- read into registers before anything is parked -- and store them behind the publish; a fifth and sixth
   * record (groups of six and more members) are stored where they are made.  What a record takes from the server, its
   * current term, is read HERE (a leader that emits records does not change its term in the same message) */
  const bool STASH = TR && RGB_X_RPC_DEFER && (PEERS_LDS || PEERS_WIDE) && lead_cls && rpcs != nullptr;
  ulonglong2 *stash = STASH ? io + lane * 8 : nullptr;
  u64 ct0 = 0;
  if (STASH) ct0 = hrow[HOT_P_TERM ^ hswz].x;
  bool done = false;
#if RGB_X_FAST
  /* the steady-state outcome of the three bulk kinds first; whoever is left takes the general clause code below */
  PairSt pst;
  pst.p = nullptr; pst.a = pst.b = make_ulonglong2(0, 0); pst.m = 0;
  if (active) {
    if (cls == 0) done = fast_aer(dev, m0, m1, m2, m3, hrow, hswz, d, &pst);
    else if (cls == 1) done = fast_aer_reply<N, TR, (PEERS_LDS || PEERS_WIDE)>(dev, m0, m1, hrow, hswz, d, prow, rrow);
    else if (cls == 2) done = fast_written(dev, m0, m1, hrow, hswz, d, &pst);
  }
  if (RGB_X_PAIR_STORE && (cls == 0 || cls == 2)) pair_store(pst, lane);       /* (wave-uniform) */
#endif
#if defined(RGB_PROFILE) && !defined(RGB_HOST_EMULATION)
  u64 tf = 0; unsigned n_fast = 0;
  if (!TR && RGB_KNOB(dev, 16u)) { tf = wall_clock64(); n_fast = (unsigned)__popcll(__ballot(done)); }   /* fast paths done */
#endif
  if (active && !done) {
#define RGB_CASE(RANK, KIND)                                                                            \
  case RANK:                                                                                            \
    RGB_MARK("begin", RANK)                                                                             \
    process_message<N, KIND, PRE, TR>(dev, m0, m1, m2, m3, base + lane, rpcs, rpc_slot_base, msg_index_base, d, tlp, \
                                      hrow, hswz, prow, rrow, stash);                                   \
    RGB_MARK("end", RANK)                                                                               \
    break;
    switch (cls) {
      RGB_CASE(0, RGB_MSG_AER) RGB_CASE(1, RGB_MSG_AER_REPLY) RGB_CASE(2, RGB_MSG_WRITTEN)
      RGB_CASE(3, RGB_MSG_APPEND) RGB_CASE(4, RGB_MSG_PIPELINE_RPCS) RGB_CASE(5, RGB_MSG_REQUEST_VOTE)
      RGB_CASE(6, RGB_MSG_VOTE_RESULT) RGB_CASE(7, RGB_MSG_AWAIT_TIMEOUT)
      RGB_CASE(8, RGB_MSG_ELECTION_TIMEOUT) RGB_CASE(9, RGB_MSG_PRE_VOTE_RPC)
      RGB_CASE(10, RGB_MSG_PRE_VOTE_RESULT) RGB_CASE(11, RGB_MSG_SNAPSHOT_WRITTEN)
      RGB_CASE(12, RGB_MSG_HEARTBEAT_RPC) RGB_CASE(13, RGB_MSG_HEARTBEAT_REPLY)
      default:
        process_message<N, RGB_MSG_CONSISTENT_QUERY, PRE, TR>(dev, m0, m1, m2, m3, base + lane, rpcs, rpc_slot_base,
                                                              msg_index_base, d, tlp, hrow, hswz, prow);
        break;
    }
#undef RGB_CASE
#if defined(RGB_PROFILE) && !defined(RGB_HOST_EMULATION)
    if (!TR && RGB_KNOB(dev, 16u)) { t2 = wall_clock64(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t2b = wall_clock64(); }
#endif
  }
  /* records parked by THIS message's clause code (the fused event below stores its own records directly and replaces
   * the decision's count: what is flushed behind the publish is what was parked, not what the merged decision says) */
  const unsigned n_parked = (STASH && active) ? (unsigned)((d.w[0] >> 48) & 0xFFull) : 0u;
  if (dev.fuse_pipeline && (cls == 1 || cls == 2)) {     /* opt-in: the pipeline_rpcs event behind the decision */
    const bool want = active && fuse_wanted(dev, d);
    if (__ballot(want) != 0ull && want) fuse_pipeline_step<N, TR>(dev, d, base + lane, rpcs, rpc_slot_base, msg_index_base);
  }
  /* the 32-byte form of the decision where its shape allows -- register arithmetic, done HERE so that in a train it
   * runs under the acknowledgements the publish step waits for (behind the publish it kept the wavefront's slot
   * for ~0.8 us longer); the storing lanes learn the record sizes from the ballot, not from the LDS slots */
  const bool is_compact = active && compact_decision(d);
  const u64 cmask = __ballot(is_compact);
  RGB_TT(4);
  if (TR) {
    /* 4. publish: every state store of this wavefront has been acknowledged by the L2 (inline assembly: the
     * compiler's wait-count pass must not drop or move it), then the advanced sequence byte of every server */
#if !defined(RGB_HOST_EMULATION) && !defined(RGB_X_TRAIN_NOWAIT)
    /* the BUILTIN, like glds_wait(): behind an inline-assembly wait the compiler's wait-count pass no longer knows
     * that the LDS-DMA copies have landed and puts a vmcnt(0) -- i.e. a wait for the previous store's acknowledgement
     * -- in front of every later LDS access: the four decision stores then leave one round trip apart (1.5 us of a
     * 10 us wavefront life).  tools/check_train_isa.py asserts the wait is in the binary, right before the byte store */
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
#endif
#if defined(RGB_X_TRAIN_NODEPS) && RGB_X_TRAIN_NODEPS == 3
    /* EXPERIMENT (with NODEPS = 1's missing poll): no sequence-byte store either -- what all of the byte traffic costs */
    if (RGB_X_TRAIN_NODEPS == 1)
#endif
#if defined(RGB_X_XOR_PUBLISH) && !defined(RGB_HOST_EMULATION)
    /* EXPERIMENT: the byte advanced by a fire-and-forget atomic XOR on its dword (old value = need, known) instead of a
     * one-byte store */
    if (has_srv && seqp != nullptr) {
      const uintptr_t a = reinterpret_cast<uintptr_t>(seqp);
      (void)__hip_atomic_fetch_xor(reinterpret_cast<u32 *>(a & ~(uintptr_t)3), ((need ^ (need + 1u)) & 0xFFu) << (8u * (unsigned)(a & 3u)),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#else
    if (has_srv && seqp != nullptr) *seqp = (unsigned char)(need + 1u);
#endif
    /* persistent train: the wavefront's NEXT row is requested now -- behind the publish (the atomic's return must not
     * sit in front of the sequence bytes) and under the decision stores; rgb_train_kernel consumes it at the top of
     * its loop.  Not earlier: a ticket held while this slice runs would start its row a whole wavefront life late,
     * and the rows that depend on it one tick later would find it uncommitted */
    if (RGB_X_TICKET_AT == 0 && ticket_ctr != nullptr) *next_ticket = rgb_take_ticket(ticket_ctr, ctl);
    /* dealt trains: this block's rotation mark, fire and forget (no value comes back, nothing waits for it; behind
     * the publish, so the sequence bytes do not wait for its acknowledgement either) */
    if (place_word != nullptr && lane == 0) {
#ifdef RGB_HOST_EMULATION
      *const_cast<u32 *>(place_word) |= place_bit;
#else
      (void)__hip_atomic_fetch_or(const_cast<u32 *>(place_word), place_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
  }
  if (STASH) {
    /* the parked rpc records, behind the publish (fire and forget, like the decisions below) */
    const unsigned n_rp = n_parked;
    if (__ballot(n_rp != 0u) != 0ull) {
      rgb_rpc *slot0 = rpcs + (size_t)(rpc_slot_base + base + lane) * (N > 1 ? N - 1 : 1);
      const u64 w0 = (u64)(msg_index_base + base + lane) | ((d.w[0] & 0xFFFFFFFFull) << 32);
#pragma unroll
      for (unsigned j = 0; j < (N > 1 ? (unsigned)N - 1u : 1u) && j < 4u; ++j) {
        if (j < n_rp) {
          const ulonglong2 a = stash[2u * j], b = stash[2u * j + 1u];
          store_rpc(slot0 + j, w0, a.x, ct0, a.y, b.x, d.w[6], b.y);
        }
      }
    }
  }
  RGB_TT(5);
  lds_barrier();      /* every lane is done with its hot row before the decisions overlay the rows */
  if (active) {
    io[lane * RGB_IO_SLOT + 0] = make_ulonglong2(d.w[0], d.w[1]);
    io[lane * RGB_IO_SLOT + 1] = make_ulonglong2(d.w[2], d.w[3]);
    if (!is_compact) {
      io[lane * RGB_IO_SLOT + 2] = make_ulonglong2(d.w[4], d.w[5]);
      io[lane * RGB_IO_SLOT + 3] = make_ulonglong2(d.w[6], d.w[7]);
    }
  }
  lds_barrier();
  if (!TR && RGB_KNOB(dev, 2u)) return true;
#ifdef RGB_X_TRAIN_NODEC      /* EXPERIMENT (breaks the output): no decision stores -- what the decision stream costs a tick */
  if (TR) return true;
#endif
  ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(dec + base);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const u32 piece = k * RGB_TICK_BLOCK + lane;
    const u32 j = piece >> 2, part = piece & 3u;
    /* decisions are never re-read on the device: non-temporal (measured -5 % per tick) */
#ifdef RGB_X_HALFDEC      /* EXPERIMENT (breaks the output): 32-byte decisions -- what would compact records be worth? */
    if (j < cnt && part < 2u) store16_nt((void *)(dst + piece), io[j * RGB_IO_SLOT + part]);
#else
    /* the upper half of a compact record is not written */
    if (j < cnt && (part < 2u || !((cmask >> j) & 1ull)))
      store16_nt((void *)(dst + piece), io[j * RGB_IO_SLOT + part]);
#endif
  }
#if defined(RGB_X_TRAIN_TIMELINE) && !defined(RGB_HOST_EMULATION)
  if (TR && dev.dbg_buf != nullptr && lane == 0 && blockIdx.x < (1u << 20)) {
    u64 *o = dev.dbg_buf + (size_t)blockIdx.x * 8;
    tt[6] = wall_clock64();
    for (int k = 0; k < 7; ++k) o[k] = tt[k];
    o[7] = (u64)(unsigned)cls | ((u64)cnt << 8) | ((u64)tt_spins << 32) | ((u64)rgb_xcc_id() << 56);
  }
#endif
#ifdef RGB_PROFILE
  if (!TR && RGB_KNOB(dev, 16u)) {
    /* run-table words read per lane: wave maximum, sum, lanes that read any */
    unsigned mx = (unsigned)tl[1], sm = (unsigned)tl[1], nz = tl[1] ? 1u : 0u;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const unsigned a = __shfl_xor(mx, o, 64), b = __shfl_xor(sm, o, 64), c = __shfl_xor(nz, o, 64);
      mx = a > mx ? a : mx; sm += b; nz += c;
    }
    if (lane == 0) {
      u64 *o = dev.dbg_buf + (size_t)blockIdx.x * 8;
      /* a wavefront whose lanes all took a fast path never stamped tl[0] / t2: zeros, not wrapped differences */
      o[0] = t0; o[1] = t1 | ((tl[0] > t1 ? tl[0] - t1 : 0) << 40);
      o[2] = t2 | ((t2b > t2 ? t2b - t2 : 0) << 40) | ((u64)cls << 60); o[3] = wall_clock64();
      o[4] = mx; o[5] = sm | ((tl[2] > t1 ? tl[2] - t1 : 0) << 24); o[6] = nz | ((tl[3] > t1 ? tl[3] - t1 : 0) << 24);
      o[7] = cnt | ((u64)n_fast << 8) | ((tf > t1 ? tf - t1 : 0) << 24);   /* lanes, lanes the fast paths took, fast paths done */
    }
  }
#endif
  return true;
}

template <int N>
__global__ __launch_bounds__(RGB_TICK_BLOCK, RGB_CLASS_MIN_WAVES(N)) void rgb_tick_classes_kernel(
    rgb_dev dev, const rgb_msg *__restrict__ msgs, rgb_tick_plan plan, const u32 *__restrict__ fam_dev,
    rgb_decision *__restrict__ dec, rgb_rpc *__restrict__ rpcs, u32 rpc_slot_base, u32 msg_index_base) {
  __shared__ ulonglong2 io[RGB_TICK_BLOCK * RGB_HOT_SLOT];   /* records (5 per slot), then hot rows (9 per slot) */
  const u32 blk_id = blockIdx.x;
  u32 q = 0, off, ncls, blk;
  if (fam_dev != nullptr) {
    /* per-family totals written by a device-side producer (2 families per class) */
    u32 n[RGB_N_CLASSES];
#pragma unroll
    for (int c = 0; c < RGB_N_CLASSES; ++c) n[c] = fam_dev[2 * c] + fam_dev[2 * c + 1];
    rgb_tick_plan p;
    rgb_make_plan(n, p, (unsigned)N);
    if (blk_id >= p.blk_end[RGB_N_CLASSES - 1]) return;
    off = p.off[0]; ncls = p.cnt[0]; blk = blk_id;
#pragma unroll
    for (int i = 1; i < RGB_N_CLASSES; ++i)
      if (blk_id >= p.blk_end[i - 1]) { q = (u32)i; off = p.off[i]; ncls = p.cnt[i]; blk = blk_id - p.blk_end[i - 1]; }
  } else {
    if (blk_id >= plan.blk_end[RGB_N_CLASSES - 1]) return;
#pragma unroll
    for (int i = 0; i < RGB_N_CLASSES - 1; ++i) q += blk_id >= plan.blk_end[i] ? 1u : 0u;
    off = plan.off[q]; ncls = plan.cnt[q];
    blk = blk_id - (q ? plan.blk_end[q - 1] : 0u);
  }
  const int cls = rgb_class_at(q);
  constexpr bool PEERS_LDS = rgb_class_slice(1, (unsigned)N) == 32u;
  const u32 SL = (PEERS_LDS && rgb_lead_class(cls)) ? 32u : (u32)RGB_TICK_BLOCK;      /* messages of this wavefront's slice */
  const u32 base = off + blk * SL;                        /* first message of this wavefront */
  const u32 end = off + ncls;
  const u32 cnt = end - base < SL ? end - base : SL;
  rgb_tick_slice<N, false>(dev, io, cls, base, cnt, SL, msgs, dec, rpcs, rpc_slot_base, msg_index_base, nullptr,
                           nullptr);
}

/* The TRAIN kernel: n_ticks consecutive ticks in ONE launch.  There is no kernel boundary between the ticks (no
 * launch gap, no dispatch ramp, no end-of-kernel write-back of the dirty L2 lines per tick, and the wavefronts of
 * neighbouring ticks drift out of phase, so one tick's memory phases run under the other's clause code); what
 * orders two messages of one server is the server's sequence byte (rgb_tick_slice).
 *
 * Coherence: the per-XCD L2s are not coherent with each other, so everything that touches a server must run on ONE
 * XCD for the life of the launch.  Servers are sharded by group (shard = group mod 8) and every tick's messages are
 * ordered by (class, shard) -- rgb_synth / the plan carry the per-shard offsets.  Within an XCD the L2 is the
 * coherence point: state stores are plain (write-through L1, line kept in the L2), state loads are L2-served.
 *
 * PERSISTENT wavefronts, placement by construction (round 4).  The grid is as many blocks as the device holds at
 * once; a block reads the id of the XCD it runs on (HW_REG_XCC_ID) and serves THAT shard for its whole life: it takes
 * the rows of its shard one by one from the shard's ticket counter (ctl) until they run out.  Nothing depends on how
 * the dispatcher deals blocks to XCDs any more (round 3 assumed round robin and checked one block in 64), every
 * wavefront slot is in use until the launch drains (the in-order round-robin dispatch of one block per slice left
 * ~15 % of the slots empty whenever one XCD was full), and a slice costs no block start.  Every lane checks that its
 * message's server belongs to the block's shard (RGB_TRAIN_ERR_ORDER otherwise: a mis-bucketed tick must not be
 * computed on another XCD's lines).
 *
 * Progress: rows are numbered tick-major per shard and tickets are handed out in order, so a wavefront only ever
 * waits for rows with SMALLER tickets -- taken by wavefronts that are running (a wavefront holds one ticket at a
 * time and takes the next only after it has published).  The poll is bounded all the same (RGB_TRAIN_ERR_SPIN), and
 * after any error the wavefronts stop taking rows (the error word travels with every ticket).
 *
 * Ticket k of shard x: tick t = the tick whose rows contain k (n_rows per tick in the plan), row = k - rows before;
 * row_tab maps the row to (class, slice of the class) -- the classes interleaved by relative position,
 * rgb_train_make_tick -- and plan[t] the (class, shard) pair to its message range.  A slice past the shard's count
 * (the fullest shard sets a class's rows) costs one more ticket. */
/* Three wavefronts per SIMD (168 registers: no spills; four need 128 and spill ~120): a train's throughput is
 * resident wavefronts / wavefront life, and the spills' scratch round trips sit on the clause code's dependency chain
 * -- 4 x 128 measured 24-25 us per tick, 3 x 168 19-20 */
#ifndef RGB_TRAIN_MIN_WAVES
#ifdef RGB_X_TRAIN_WAVES            /* (a plain number on the command line: tools/build_variants.sh) */
#define RGB_TRAIN_MIN_WAVES(N) RGB_X_TRAIN_WAVES
#else
#define RGB_TRAIN_MIN_WAVES(N) 3
#endif
#endif
/* the PERSISTENT form of groups of eight members: 167 registers + 2 spilled (12 bytes of scratch) at three wavefronts per
 * SIMD; its 16 KiB of LDS per wavefront admit ten per compute unit anyway, so it is compiled for two per SIMD (eight per
 * unit) and no scratch -- every hot kernel of every group size is scratch-free (tests/test_kernel_resources.py) */
#ifndef RGB_TRAIN_PERSIST_MIN_WAVES
#define RGB_TRAIN_PERSIST_MIN_WAVES(N) ((N) >= 8 ? 2 : RGB_TRAIN_MIN_WAVES(N))
#endif
#define RGB_TRAIN_CTL_ARRIVE 8u     /* ctl words 8..15: blocks arrived per XCC (devices with fewer XCCs than shards) */
#define RGB_TRAIN_CTL_TICKET 32u    /* ctl word 32 (1 + x): next row of shard x (one 128-byte line per shard)         */
/* the kernel's one argument.  The persistent loop re-reads it from the kernarg segment through a pointer the compiler
 * cannot see through, every iteration: hoisted out of the loop, the fields (and everything derived from them and
 * from the lane id, for fifteen class paths) stay live across the whole slice -- 370 spilled SGPRs and 209 spilled
 * VGPRs at first try */
struct rgb_train_args {
  rgb_dev dev;
  const rgb_msg *msgs;
  const unsigned char *stamps;
  const rgb_train_tick *plan;
  const u32 *row_tab;
  rgb_decision *dec;
  rgb_rpc *rpcs;
  u32 *ctl;
  u32 tick_stride, rpt, n_ticks, rpc_ring, index_base, n_xcc;
  u32 tab_rpt;                        /* rows per tick of row_tab (>= rpt, the rows per tick of the dealt GRID: a device-built
                                         plan's table is sized by the rows bound, its grid by the rows its ticks have when
                                         the host has been told -- rgb_train_plan_fit) */
  const unsigned char *snap_stamps;   /* snapshots inside the launch: the bytes every server must show, laid out like  */
  rgb_leaderboard_row *snap_rows;     /* dev.seq, one array per snapshot ordinal; the rows, n_groups per ordinal        */
};

/* One row of an in-launch leaderboard snapshot (plan class RGB_PC_SNAP): lane = one group of shard x.  Same protocol
 * as a message slice -- poll the members' sequence bytes, read (L2-served), publish the advanced bytes once the loads
 * have returned -- with nothing to commit: the row is rgb_leaderboard_kernel's, group by group at its own boundary. */
template <int N>
__device__ __forceinline__ bool rgb_train_snap_slice(const rgb_dev &dev, u32 x, u32 j, u32 lane,
                                                     const unsigned char *__restrict__ stamps,
                                                     rgb_leaderboard_row *__restrict__ rows, u32 *__restrict__ ctl) {
  const u32 G = dev.n_servers / (u32)N;
  const u32 gi = j * RGB_TICK_BLOCK + lane;                /* group of the shard */
  const u32 g = gi * RGB_TRAIN_SHARDS + x;
  const bool live = g < G;
  const u32 sbase = live ? rgb_seq_index(g * (u32)N, (unsigned)N, dev.seq_stride) : 0u;   /* the members' bytes follow */
  unsigned need[N];
#pragma unroll
  for (int m = 0; m < N; ++m) need[m] = live ? (unsigned)stamps[sbase + m * RGB_SEQ_SPREAD] : 0u;
  unsigned spins = 0;
  bool late = live;
  for (;;) {
    if (late) {
      bool ok = true;
#pragma unroll
      for (int m = 0; m < N; ++m) {
#ifdef RGB_HOST_EMULATION
        const unsigned cur = dev.seq[sbase + m * RGB_SEQ_SPREAD];
#else
        const unsigned cur = __hip_atomic_load(dev.seq + sbase + m * RGB_SEQ_SPREAD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        ok = ok && cur == need[m];
      }
      late = !ok;
    }
    if (__ballot(late) == 0ull) break;
    spins += 1;
    bool give_up = spins > RGB_TRAIN_SPIN_LIMIT;
#ifndef RGB_HOST_EMULATION
    if ((spins & 15u) == 0u && __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) give_up = true;
    __builtin_amdgcn_s_sleep(RGB_TRAIN_SLEEP);
    if (spins > 64u) __builtin_amdgcn_s_sleep(127);
#endif
    if (give_up) {
      if (lane == 0) atomicOr(ctl, (u32)RGB_TRAIN_ERR_SPIN);
      return false;
    }
  }
  if (live) {
    u32 leader = RGB_NONE, n_leaders = 0;
    u64 term = 0, lead_term = 0, ci = 0, la = 0, max_ci = 0, max_la = 0;
#pragma unroll
    for (int m = 0; m < N; ++m) {
      const ulonglong2 *hp = reinterpret_cast<const ulonglong2 *>(dev.hot + ((size_t)g * N + m) * RGB_HOT_WORDS);
      const ulonglong2 h0 = ldg16(true, hp + HOT_P_TERM), h1 = ldg16(true, hp + HOT_P_CI);
      const u64 ct = h0.x, pk = h0.y;
      if (ct > term) term = ct;
      if (h1.x > max_ci) max_ci = h1.x;
      if (h1.y > max_la) max_la = h1.y;
      if (pk_get(pk, PK_ROLE_SH, 3) == RGB_ROLE_LEADER) {
        n_leaders++;
        if (leader == RGB_NONE || ct > lead_term) { leader = (u32)m; lead_term = ct; ci = h1.x; la = h1.y; }
      }
    }
    rgb_leaderboard_row r;
    r.leader = leader; r.n_leaders = n_leaders; r.term = term;
    r.commit_index = leader == RGB_NONE ? max_ci : ci;
    r.last_applied = leader == RGB_NONE ? max_la : la;
    rows[g] = r;
  }
#ifndef RGB_HOST_EMULATION
  __builtin_amdgcn_s_waitcnt(0x0F70);     /* every row load has returned (vmcnt 0) before the bytes move */
  asm volatile("" ::: "memory");
#endif
  if (live) {
#pragma unroll
    for (int m = 0; m < N; ++m) dev.seq[sbase + m * RGB_SEQ_SPREAD] = (unsigned char)(need[m] + 1u);
  }
  return true;
}
/* DEALT trains (one block per row, block b serves shard b mod 8): the round-3 dispatch, which needs the dispatcher to
 * deal the blocks of a launch round robin over the XCDs -- block b on XCD (b + r) mod 8, r fixed per launch (what
 * /opt/skills/guides/cdna_hip_programming.md, "XCD-aware blockIdx swizzle", describes as the default and every
 * XCD-aware tiling relies on).  It is still VERIFIED, in every launch and for every block: (1) a context only uses
 * this form when its calibration launch was dealt that way; (2) every block ORs the bit of its rotation
 * (XCC id - shard) mod 8 into one of RGB_TRAIN_MARK_WORDS control words (a fire-and-forget atomic behind its publish:
 * no value returns, nothing waits -- a returning atomic or an agent-scope load per block cost 1-5 % of the tick), and
 * rgb_train_prolog_kernel -- in front of the NEXT launch on the same control words, and behind the last one when the
 * host asks (rgb_train_status, rgb_submit) -- raises RGB_TRAIN_ERR_PLACEMENT unless all marks of the launch are ONE bit.  rgb_submit then repairs the batch (undo log + one launch per round) and the context goes over
 * to the persistent form for good.  Measured 2-7 % faster than the persistent form on the 65 536 x 5 closed loop (a
 * ticket and its row look-up stand ~2 us in front of every slice, and the device is not short of wavefront slots),
 * which is why it is the default where it holds. */
#define RGB_TRAIN_MARK_WORDS 64u    /* ctl words 320 + 32 j (one 128-byte line each), j = row mod 64 */
#define RGB_TRAIN_CTL_MARK 320u
template <int N>
__global__ __launch_bounds__(RGB_TICK_BLOCK, RGB_TRAIN_MIN_WAVES(N)) void rgb_train_dealt_kernel(rgb_train_args args) {
  __shared__ ulonglong2 io[!RGB_TRAIN_RUNS_LDS ? RGB_TICK_BLOCK * RGB_HOT_SLOT
                           : rgb_class_slice(1, (unsigned)N) == 32u ? 12 * RGB_TICK_BLOCK : 16 * RGB_TICK_BLOCK];
  const u32 x = blockIdx.x & (RGB_TRAIN_SHARDS - 1u), k = blockIdx.x / RGB_TRAIN_SHARDS;
  const u32 rpt = args.rpt, n_ticks = args.n_ticks;
  const u32 t = k / rpt, row = k - t * rpt;
  /* The block's prologue is a chain of scalar-memory round trips in front of the first record load, and a wavefront
   * slot is held all the while (the launch is bound by wave-time over slots): the tick's header and the row's table
   * entry are requested TOGETHER, from clamped addresses, and looked at once both are here -- written with an early
   * return between them they were two round trips (and the kernel arguments behind each return a third and fourth) */
  const u32 tc = t < n_ticks ? t : n_ticks - 1u;           /* (a launch has at least one tick) */
  const rgb_train_tick *p = args.plan + tc;
  const u32 n_rows = p->n_rows;
  const u32 e = args.row_tab[(size_t)tc * args.tab_rpt + row];
  /* (bitwise: both loads are wanted by the one decision; no table entry is all ones -- its class byte is <= 30) */
  if ((t >= n_ticks) | (row >= n_rows) | (e == 0xFFFFFFFFu)) return;
  const u32 pc = e >> 24;                                  /* plan class = 2 x class + sub-bucket */
  if (pc == RGB_PC_SNAP) {
    /* the snapshot in front of this tick; in front of the launch's FIRST tick it is the caller's (outside the launch) */
    if (t != 0u && p->snap != 0u && args.snap_rows != nullptr)
      (void)rgb_train_snap_slice<N>(args.dev, x, e & 0xFFFFFFu, threadIdx.x,
                                    args.snap_stamps + (size_t)(p->snap - 1u) * args.dev.seq_stride * RGB_TRAIN_SHARDS,
                                    args.snap_rows + (size_t)(p->snap - 1u) * (args.dev.n_servers / (u32)N), args.ctl);
    return;
  }
  const int cls = (int)(pc >> 1);
  const u32 off = p->off[pc][x], ncls = p->cnt[pc][x];
  const u32 SL = rgb_train_class_slice(cls, (unsigned)N);
  const u32 lbase = (e & 0xFFFFFFu) * SL;
  if (lbase >= ncls) return;
  const u32 cnt = ncls - lbase < SL ? ncls - lbase : SL;
  const u32 tick_stride = args.tick_stride;
  const size_t toff = tick_stride ? (size_t)t * tick_stride : (size_t)p->msg_base;
  rgb_rpc *rp = args.rpcs ? args.rpcs + (tick_stride ? (size_t)(t % args.rpc_ring) * tick_stride : toff) * (N > 1 ? N - 1 : 1) : nullptr;
  u32 unused = 0;
  (void)rgb_tick_slice<N, true>(args.dev, io, cls, off + lbase, cnt, SL, args.msgs + toff, args.dec + toff, rp, 0,
                                args.index_base + (u32)toff, args.ctl, args.stamps + toff, nullptr, &unused, x,
                                threadIdx.x, args.ctl + RGB_TRAIN_CTL_MARK + 32u * (k & (RGB_TRAIN_MARK_WORDS - 1u)),
                                1u << ((rgb_xcc_id() - x) & (RGB_TRAIN_SHARDS - 1u)));
}

/* BEHIND every train launch (rgb_launch_train): the rotation marks the dealt launch left on these control words must
 * be one bit -- all of them have landed, the kernels are ordered by the stream -- else RGB_TRAIN_ERR_PLACEMENT (sticky,
 * word 0); then every per-launch word is cleared for the next launch.  clear = 0: verify only. */
__global__ void rgb_train_prolog_kernel(u32 *__restrict__ ctl, u32 clear) {
  if (threadIdx.x < RGB_TRAIN_MARK_WORDS) {
    u32 v = ctl[RGB_TRAIN_CTL_MARK + 32u * threadIdx.x];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v |= __shfl_xor(v, o, 64);
    if (threadIdx.x == 0 && (v & (v - 1u)) != 0u) atomicOr(ctl, (u32)RGB_TRAIN_ERR_PLACEMENT);
  }
  __syncthreads();
  if (clear)
    for (u32 w = 1u + threadIdx.x; w < RGB_TRAIN_CTL_WORDS; w += blockDim.x) ctl[w] = 0u;
}

template <int N>
__global__ __launch_bounds__(RGB_TICK_BLOCK, RGB_TRAIN_PERSIST_MIN_WAVES(N)) void rgb_train_kernel(rgb_train_args args) {
  /* records, then hot rows; a leader-side slice: 32 hot rows | 32 peers rows | the first line of 32 run tables
   * (12 KiB: twelve wavefronts per CU -- three per SIMD, what the registers allow -- hold 144 of the 160 KiB) */
  __shared__ ulonglong2 io[!RGB_TRAIN_RUNS_LDS ? RGB_TICK_BLOCK * RGB_HOT_SLOT
                           : rgb_class_slice(1, (unsigned)N) == 32u ? 12 * RGB_TICK_BLOCK : 16 * RGB_TICK_BLOCK];
  /* the shard this block serves: the XCD it runs on */
  u32 x;
  {
    const u32 n_xcc = args.n_xcc;
    const u32 xcc = rgb_xcc_id() & (RGB_TRAIN_SHARDS - 1u);
    if (n_xcc >= RGB_TRAIN_SHARDS) x = xcc;
    else {
      /* fewer XCCs than shards (a partitioned device; the CPU emulation has one): the blocks of an XCC take its
       * shards xcc, xcc + n_xcc, .. in arrival order -- the grid holds at least 8 / n_xcc blocks per XCC */
      u32 a = 0;
      if (threadIdx.x == 0) a = atomicAdd(args.ctl + RGB_TRAIN_CTL_ARRIVE + (xcc & (n_xcc - 1u)), 1u);
      a = rgb_ticket_value(a);
      x = (xcc & (n_xcc - 1u)) + n_xcc * (a % (RGB_TRAIN_SHARDS / n_xcc));
    }
  }
  u32 raw = rgb_take_ticket(args.ctl + RGB_TRAIN_CTL_TICKET * (1u + x), args.ctl);
  u32 t = 0, cum = 0;
  for (;;) {
#if defined(RGB_HOST_EMULATION) || !defined(__HIP_DEVICE_COMPILE__)   /* (the host pass only parses the kernel) */
    const rgb_train_args *A = &args;
    u32 lane = threadIdx.x;
#else
    /* the constant address space stays on the pointer: loads through it are invariant scalar loads (through a generic
     * pointer every state store could alias the arguments and each use would re-load them behind a wait) */
    typedef const __attribute__((address_space(4))) rgb_train_args *args_ptr;
    args_ptr A = (args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(A));
    u32 lane = threadIdx.x;
    asm volatile("" : "+v"(lane));
#endif
    const rgb_dev dev = A->dev;
    const u32 k = rgb_ticket_value(raw);
    if (rgb_ticket_err(raw) != 0u) break;                 /* the launch has failed: drain */
    /* the tick of row k (tickets only grow: t and the rows before it are carried along) */
    const rgb_train_tick *plan = A->plan;
    const u32 n_ticks = A->n_ticks;
    u32 nr = 0;
    while (t < n_ticks) {
      nr = plan[t].n_rows;
      if (k - cum < nr) break;
      cum += nr; ++t;
    }
    if (t >= n_ticks) break;
    const rgb_train_tick *p = plan + t;
    const u32 e = A->row_tab[(size_t)t * A->tab_rpt + (k - cum)];
    const u32 pc = e >> 24;                                /* plan class = 2 x class + sub-bucket */
    u32 *const tk = A->ctl + RGB_TRAIN_CTL_TICKET * (1u + x);
    if (pc == RGB_PC_SNAP) {
      if (t != 0u && p->snap != 0u && A->snap_rows != nullptr) {
        if (!rgb_train_snap_slice<N>(dev, x, e & 0xFFFFFFu, lane,
                                     A->snap_stamps + (size_t)(p->snap - 1u) * dev.seq_stride * RGB_TRAIN_SHARDS,
                                     A->snap_rows + (size_t)(p->snap - 1u) * (dev.n_servers / (u32)N), A->ctl))
          break;
      }
      raw = rgb_take_ticket(tk, A->ctl);
      continue;
    }
    const int cls = (int)(pc >> 1);
    const u32 off = p->off[pc][x], ncls = p->cnt[pc][x];
    const u32 SL = rgb_train_class_slice(cls, (unsigned)N);
    const u32 lbase = (e & 0xFFFFFFu) * SL;
    if (lbase >= ncls) { raw = rgb_take_ticket(tk, A->ctl); continue; }
    const u32 cnt = ncls - lbase < SL ? ncls - lbase : SL;
    /* ticks a fixed stride apart with a ring of rpc regions (device-resident streams), or -- tick_stride = 0 -- packed
     * one behind the other, every message owning the rpc slots of its index in the whole buffer (rgb_submit's rounds) */
    const u32 tick_stride = A->tick_stride;
    const size_t toff = tick_stride ? (size_t)t * tick_stride : (size_t)p->msg_base;
    rgb_rpc *rp = A->rpcs ? A->rpcs + (tick_stride ? (size_t)(t % A->rpc_ring) * tick_stride : toff) * (N > 1 ? N - 1 : 1) : nullptr;
    lds_barrier();                                        /* the previous slice is done with the staging area */
    raw = 0;
    if (!rgb_tick_slice<N, true>(dev, io, cls, off + lbase, cnt, SL, A->msgs + toff, A->dec + toff, rp, 0,
                                 A->index_base + (u32)toff, A->ctl, A->stamps + toff, tk, &raw, x, lane))
      break;
  }
}

/* once per context: out[0] |= 1 << (XCC id) over one launch that fills the device: the set of XCC ids blocks run on.
 * The host accepts ids 0 .. n-1 with n = 1, 2, 4 or 8 (a block serves the shards congruent to its XCC id mod n) */
/* out[1] |= 1 << ((XCC id - blockIdx) mod 8): one bit = the dispatcher dealt this launch round robin (what the
 * dealt form of a train needs, and checks again in every block of every launch) */
__global__ void rgb_train_calibrate_kernel(u32 *__restrict__ out) {
  if (threadIdx.x == 0) {
    const u32 xcc = rgb_xcc_id();
    atomicOr(out, 1u << xcc);
    atomicOr(out + 1, 1u << ((xcc - blockIdx.x) & (RGB_TRAIN_SHARDS - 1u)));
  }
}

/* Sequence stamps of a train: seq_cnt[i] = the value server (at sequence index i) will hold when the next message
 * reaches it -- a copy of dev.seq taken by the launcher before the first tick.  Tick by tick in train order, every
 * message takes its server's count as its stamp and advances it (one message per server per tick: no two lanes
 * share a counter). */
__global__ void rgb_train_seq_kernel(rgb_dev dev, const rgb_msg *__restrict__ msgs, u32 n, unsigned char *__restrict__ seq_cnt,
                                     unsigned char *__restrict__ stamps) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 w = *reinterpret_cast<const u64 *>(msgs + i);
  const u32 sv = (u32)(w & 0xFFFFFFFFull);
  if (((w >> 32) & 0xFFull) == RGB_MSG_NOP || sv >= dev.n_servers) { stamps[i] = 0; return; }
  const u32 k = rgb_seq_index(sv, dev.n_members, dev.seq_stride);
  const unsigned char c = seq_cnt[k];
  seq_cnt[k] = (unsigned char)(c + 1u);
  stamps[i] = c;
}

/* ------------------------------------------------------------ synthetic load ---- */
/* Device-side load generator (include/ra_gpu_batch_synth.h).  One lane per GROUP reads the hot
 * lines of its N members (and the leader's peers row) and synthesises this tick's messages.  The
 * tick comes out COMPACTED and ordered by clause family (kind, success flag), the order
 * rgb_submit gives host batches:  pass 1 (rgb_synth_kernel<N,false>) counts messages per family,
 * pass 2 (<N,true>) recomputes them (same counter-based PRNG) and writes each one at
 * family_base + block reservation + rank. */

#define SYN_FAMILIES RGB_N_FAMILIES

__device__ __forceinline__ u64 sm64(u64 &x) {
  x += 0x9E3779B97F4A7C15ull;
  u64 z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

struct SynMember { u64 ct, ci, la, li, lt, lwi, lwt, pk, first, lrs, lrt, prs, prt; };

/* a Lane good for fetch_term against member x's log */
template <class Lane>
__device__ __forceinline__ void syn_lane(Lane &T, const SynMember &x, const u64 *runs) {
  T.first = x.first; T.li = x.li; T.lrs = x.lrs; T.lrt = x.lrt; T.prs = x.prs; T.prt = x.prt;
  T.push_cnt = 0; T.n_runs = (unsigned)pk_get(x.pk, PK_NRUNS_SH, 5);
  T.runs = runs; T.runs_lds = nullptr;
#ifdef RGB_PROFILE
  T.prof_noprobe = false; T.prof_nloads = 0;
#endif
}

struct SynMsg {
  u32 server; unsigned kind, from, flags, gap; u64 term, a, b, c; u32 n_entries, n_run0; u64 run0, run1;
  bool off_steady;      /* the producer's bucketing hint (rgb_bucket_hinted): never part of the record */
};

__device__ __forceinline__ void syn_store(rgb_msg *slot, const SynMsg &m) {
  ulonglong2 *o = reinterpret_cast<ulonglong2 *>(slot);
  u64 w0 = (u64)m.server | ((u64)(m.kind & 0xFF) << 32) | ((u64)(m.from & 0xFF) << 40) |
           ((u64)(m.flags & 0xFF) << 48) | ((u64)(m.gap & 0xFF) << 56);
  o[0] = make_ulonglong2(w0, m.term);
  o[1] = make_ulonglong2(m.a, m.b);
  o[2] = make_ulonglong2(m.c, (u64)m.n_entries | ((u64)m.n_run0 << 32));
  o[3] = make_ulonglong2(m.run0, m.run1);
}

__device__ __forceinline__ SynMsg syn_msg(u32 server, unsigned kind, unsigned from, unsigned flags, u64 term,
                                          u64 a, u64 b, u64 c, u32 n_entries = 0, u32 n_run0 = 0, u64 run0 = 0) {
  SynMsg m;
  m.server = server; m.kind = kind; m.from = from; m.flags = flags; m.gap = 0; m.term = term;
  m.a = a; m.b = b; m.c = c; m.n_entries = n_entries; m.n_run0 = n_run0; m.run0 = run0; m.run1 = 0;
  m.off_steady = false;
  return m;
}

/* the messages of group g for this tick; emit(msg) is called once per message */
template <int N, class Emit>
__device__ __forceinline__ void synth_group(const rgb_dev &dev, u64 seed, u64 tick, u32 g, Emit &&emit) {
  u64 rs = seed ^ (tick * 0xD1B54A32D192ED03ull) ^ ((u64)g * 0x9E3779B97F4A7C15ull);
  SynMember mb[N];
  int leader = -1, cand = -1, prev = -1, hi = 0;
  u64 lead_ct = 0, maxct = 0;
#pragma unroll
  for (int m = 0; m < N; ++m) {
    const ulonglong2 *hp = reinterpret_cast<const ulonglong2 *>(dev.hot + ((size_t)g * N + m) * RGB_HOT_WORDS);
    const ulonglong2 h0 = hp[HOT_P_TERM], h1 = hp[HOT_P_CI], h2 = hp[HOT_P_LI], h3 = hp[HOT_P_LW], h5 = hp[HOT_P_FIRST],
                     h6 = hp[HOT_P_LRT], h7 = hp[HOT_P_PEND];
    mb[m].ct = h0.x; mb[m].pk = h0.y; mb[m].ci = h1.x; mb[m].la = h1.y; mb[m].li = h2.x;
    mb[m].lt = h2.y; mb[m].lwi = h3.x; mb[m].lwt = h3.y; mb[m].first = h5.x; mb[m].lrs = h5.y;
    mb[m].lrt = h6.x; mb[m].prs = h6.y; mb[m].prt = h7.x;
  }
#pragma unroll
  for (int m = 0; m < N; ++m) {
    const unsigned role = (unsigned)pk_get(mb[m].pk, PK_ROLE_SH, 3);
    if (mb[m].ct > maxct) maxct = mb[m].ct;
    if (role == RGB_ROLE_LEADER && (leader < 0 || mb[m].ct > lead_ct)) { leader = m; lead_ct = mb[m].ct; }
  }
#pragma unroll
  for (int m = 0; m < N; ++m) {
    const unsigned role = (unsigned)pk_get(mb[m].pk, PK_ROLE_SH, 3);
    if (role == RGB_ROLE_CANDIDATE && mb[m].ct == maxct && cand < 0) cand = m;
    if (role == RGB_ROLE_PRE_VOTE && mb[m].ct == maxct && prev < 0) prev = m;
    /* most advanced member: highest term, then most up-to-date log */
    const bool better = mb[m].ct > mb[hi].ct ||
                        (mb[m].ct == mb[hi].ct && (mb[m].lt > mb[hi].lt ||
                                                   (mb[m].lt == mb[hi].lt && mb[m].li > mb[hi].li)));
    if (better) hi = m;
  }
  auto sid = [&](int m) -> u32 { return g * N + (u32)m; };
  auto other = [&](int m, u64 r) -> int { return N > 1 ? (int)((m + 1 + r % (N - 1)) % N) : m; };

  const bool healthy = leader >= 0 && lead_ct == maxct;
  if (!healthy) {
    /* election traffic only */
    const u64 r = sm64(rs);
    if (leader >= 0) {
      /* a member is ahead of the leader: its failed reply deposes the leader (a7) */
      int mh = hi == leader ? other(leader, r) : hi;
      emit(syn_msg(sid(leader), RGB_MSG_AER_REPLY, mh, 0, mb[mh].ct, mb[mh].li + 1, mb[mh].lwi, mb[mh].lwt));
    } else if (cand >= 0) {
      emit(syn_msg(sid(cand), RGB_MSG_VOTE_RESULT, other(cand, r), RGB_MF_SUCCESS, mb[cand].ct, 0, 0, 0));
    } else if (prev >= 0) {
      emit(syn_msg(sid(prev), RGB_MSG_PRE_VOTE_RESULT, other(prev, r), RGB_MF_SUCCESS, mb[prev].ct, 0, 0,
                   (dev.qry + (size_t)sid(prev) * RGB_QRY_WORDS)[QRY_TOKEN]));
    } else {
      emit(syn_msg(sid(hi), RGB_MSG_ELECTION_TIMEOUT, RGB_NONE, 0, 0, 0, 0, (r | 1ull) & 0xFFFFFFFFFFFFull));
    }
    return;
  }
  const int l = leader;
  const SynMember &ld = mb[l];
  bool used[N];
#pragma unroll
  for (int m = 0; m < N; ++m) used[m] = false;
  /* term churn: 5 % of the groups, one member gets request_vote term+1 */
  const u64 rc = sm64(rs);
  if (rc % 100 < 5 && !RGB_KNOB(dev, 128u)) {
    const int churn = (int)((rc >> 8) % N);
    const u64 r2 = sm64(rs);
    const u64 lli = mb[churn].li + (r2 % 3) > 0 ? mb[churn].li + (r2 % 3) - 1 : 0;
    emit(syn_msg(sid(churn), RGB_MSG_REQUEST_VOTE, other(churn, r2 >> 8), 0, mb[churn].ct + 1, lli,
                 mb[churn].lt, 0));
    used[churn] = true;
  }
  /* ---- log compaction: a member whose log spans many terms gets the snapshot_written event of a
   * snapshot taken at its last_applied index (release_cursor), which releases the old term runs ---- */
#pragma unroll
  for (int m = 0; m < N; ++m) {
    if (used[m]) continue;
    const SynMember &x = mb[m];
    const unsigned nr = (unsigned)pk_get(x.pk, PK_NRUNS_SH, 5);
    if (nr < 4 || !(x.first <= x.li) || x.la < x.first || x.la > x.li) continue;
    LaneT<false> T;
    syn_lane(T, x, dev.runs + (size_t)sid(m) * dev.max_runs * 2);
    const u64 t = fetch_term(T, x.la);
    if (t == UNDEF) continue;
    emit(syn_msg(sid(m), RGB_MSG_SNAPSHOT_WRITTEN, RGB_NONE, 0, 0, x.la, t, 0));
    used[m] = true;
  }
  /* ---- consistent-query heartbeats in flight: the leader's query_index and the first peer that
   * has not confirmed it (the row is only read once a query was ever issued) ---- */
  u64 lq = 0; int hb_lag = -1; unsigned hb_behind = 0;
  if (pk_get(ld.pk, PK_QSELF_SH, 1)) {
    const u64 *q = dev.qry + (size_t)sid(l) * RGB_QRY_WORDS;
    lq = q[0];
    const bool peers_nz = pk_get(ld.pk, PK_QPEER_SH, 1) != 0;
#pragma unroll
    for (int m = 0; m < N; ++m) {
      if (m == l) continue;
      const u64 qm = peers_nz ? q[1 + m] : 0;
      if (qm < lq) { hb_behind |= 1u << m; if (hb_lag < 0) hb_lag = m; }
    }
  }
  /* ---- leader-side message ---- */
  if (!used[l]) {
    const u64 r = sm64(rs), r2 = sm64(rs);
    const bool nonempty = ld.first <= ld.li;
    if (ld.lt != ld.ct || !nonempty) {
      /* just elected: the noop of the new term, pipelined with Force */
      emit(syn_msg(sid(l), RGB_MSG_APPEND, RGB_NONE, RGB_MF_FORCE, 0, 0, 0, 0, 1));
    } else if (ld.lwi < ld.li && (r & 3) == 0) {
      const u64 a = ld.lwi + 1 > ld.first ? ld.lwi + 1 : ld.first;
      SynMsg w = syn_msg(sid(l), RGB_MSG_WRITTEN, RGB_NONE, 0, ld.lt, a, ld.li, 0);
      w.off_steady = RGB_X_HINT && dev.synth_hint >= 1u;   /* the owner is in state leader (ra_server_proc knows) */
      emit(w);
    } else if (hb_lag >= 0 && (r >> 40) % 3 == 0) {
      /* a follower that has not confirmed the current query index answers the heartbeat */
      emit(syn_msg(sid(l), RGB_MSG_HEARTBEAT_REPLY, hb_lag, 0, ld.ct, lq, 0, 0));
    } else if ((r >> 44) % 50 == 0) {
      emit(syn_msg(sid(l), RGB_MSG_CONSISTENT_QUERY, RGB_NONE, 0, 0, 0, 0, 0));     /* 2 % */
    } else {
      unsigned v = (unsigned)((r >> 8) % 100);
      if (RGB_KNOB(dev, 128u) && v >= 75 && v < 80) v = 0;   /* profiling: no failed replies */
      const int j = other(l, r >> 16);
      const u64 *pr = dev.peers + (size_t)sid(l) * dev.peer_stride;
      const u64 mi = pr[PEER_MI(j, N)];
      LaneT<false> T;                                   /* term lookups against the leader's log */
      syn_lane(T, ld, dev.runs + (size_t)sid(l) * dev.max_runs * 2);
      if (v < 75) {
        /* what follower j has durably written (its first reply after an election jumps the
         * leader's match_index from 0 to there), else a small step past the known match */
        u64 last = mi + r2 % 5;
        if (mb[j].lwi > last) last = mb[j].lwi;
        if (last > ld.li) last = ld.li;
        u64 nxt = last + 1 + (r2 >> 8) % 3; if (nxt > ld.li + 1) nxt = ld.li + 1;
        u64 t = fetch_term(T, last); if (t == UNDEF) t = 0;
        emit(syn_msg(sid(l), RGB_MSG_AER_REPLY, j, RGB_MF_SUCCESS, ld.ct, nxt, last, t));
      } else if (v < 80) {
        u64 lo = ld.first > 0 ? ld.first - 1 : 0;
        u64 last = mi > r2 % 4 ? mi - r2 % 4 : 0; if (last < lo) last = lo;
        u64 t = fetch_term(T, last); if (t == UNDEF) t = 0;
        if ((r2 >> 8) & 1) t += 1;                         /* half of them with a wrong term */
        emit(syn_msg(sid(l), RGB_MSG_AER_REPLY, j, 0, ld.ct, last + 1, last, t));
      } else {
        emit(syn_msg(sid(l), RGB_MSG_APPEND, RGB_NONE, 0, 0, 0, 0, 0, 1 + (u32)(r2 % 4)));
      }
    }
  }
  /* ---- follower-side messages ---- */
#pragma unroll
  for (int j = 0; j < N; ++j) {
    if (j == l || used[j]) continue;
    const SynMember &f = mb[j];
    const u64 r = sm64(rs), r2 = sm64(rs);
    if (((hb_behind >> j) & 1u) && (r >> 40) % 8 == 0) {
      emit(syn_msg(sid(j), RGB_MSG_HEARTBEAT_RPC, l, 0, ld.ct, lq, 0, 0));
    } else if ((r & 1) == 0) {
      unsigned v = (unsigned)((r >> 8) % 100);
      if (RGB_KNOB(dev, 128u)) v = 0;                           /* profiling: tail appends only */
      const u64 eterm = ld.ct > f.lt ? ld.ct : f.lt;
      u64 prev_i = f.li, prev_t = f.lt, run0 = eterm;
      u32 n_ent = 1 + (u32)(r2 % 8);
      if (v >= 80 && v < 85) { n_ent = 0; }                                   /* heartbeat */
      else if (v >= 85 && v < 90) { prev_i = f.li + 1 + (r2 >> 8) % 3; prev_t = ld.ct; n_ent = (u32)((r2 >> 16) % 3); }
      else if (v >= 90 && v < 95) { prev_t = f.lt + 1; n_ent = 0; }           /* term mismatch */
      else if (v >= 95) {
        const u64 back = 1 + (r2 >> 8) % 4;                                   /* overlapping resend */
        u64 floor_i = f.lrs > f.la ? f.lrs : f.la; if (f.first > floor_i) floor_i = f.first;
        if (f.first <= f.li && f.li >= back && f.li - back >= floor_i) {
          prev_i = f.li - back; prev_t = f.lt; n_ent = (u32)back; run0 = f.lt;
        }
      }
      SynMsg w = syn_msg(sid(j), RGB_MSG_AER, l, 0, ld.ct, prev_i, prev_t, ld.ci, n_ent, n_ent, run0);
      /* the receiver's own state name; level 2: + what its owner sees by comparing the rpc's header with fields it
       * holds (current_term, leader_id, ra_log's last index and term: src/ra_server.erl:1283-1303 reads exactly
       * these) -- not a plain append at the tail from the leader it knows, in the term it is in */
      bool off = pk_get(f.pk, PK_ROLE_SH, 3) != RGB_ROLE_FOLLOWER;
      if (dev.synth_hint >= 2u)
        off = off || f.ct != ld.ct || (unsigned)pk_get(f.pk, PK_LEADER_SH, 4) != (unsigned)l || !(f.first <= f.li) ||
              prev_i != f.li || prev_t != f.lt || (n_ent != 0 && run0 != f.lt);
      w.off_steady = RGB_X_HINT && dev.synth_hint >= 1u && off;
      emit(w);
    } else if (f.lwi < f.li && f.first <= f.li && (r >> 8) % 10 < 8) {
      const u64 a = f.lwi + 1 > f.first ? f.lwi + 1 : f.first;
      SynMsg w = syn_msg(sid(j), RGB_MSG_WRITTEN, RGB_NONE, 0, f.lt, a, f.li, 0);
      bool off = pk_get(f.pk, PK_ROLE_SH, 3) != RGB_ROLE_FOLLOWER;
      if (dev.synth_hint >= 2u) off = off || (unsigned)pk_get(f.pk, PK_LEADER_SH, 4) == SLOT_NONE4;   /* nobody to reply to */
      w.off_steady = RGB_X_HINT && dev.synth_hint >= 1u && off;
      emit(w);
    }
  }
}

/* The tick comes out ordered by BUCKET = (class of the message kind, shard of the group, success flag): class-major,
 * so every kernel class is one contiguous range (what the per-tick class kernel needs; inside it the failed and the
 * successful replies still sit in runs), and inside a class the messages of one shard (group mod RGB_TRAIN_SHARDS)
 * are contiguous -- what a train launch needs (rgb_train_kernel).  Inside a bucket the messages are in GROUP order
 * (by generator block = 64 consecutive groups), the same in every tick: a server's messages sit at the same relative
 * position of their buckets tick after tick, which is what lets a train's ticks follow each other at a constant lag.
 * Scratch words (zeroed by the launcher per tick): fam_total[RGB_N_FAMILIES] | bkt_total[RGB_N_BUCKETS] |
 * bkt_base[RGB_N_BUCKETS] | blk_cnt[blocks][RGB_N_BUCKETS] (pass 1: per-block counts; after the scan: per-block bases).
 * Pass 1 (WRITE = false) counts per block and bucket, rgb_synth_scan_kernel turns the counts into bases (and the
 * family totals, the per-kind counts and the tick's size), pass 2 recomputes the messages (same counter-based PRNG)
 * and writes each one at its block's base + rank. */
template <int N, bool WRITE>
__global__ __launch_bounds__(64) void rgb_synth_kernel(rgb_dev dev, u64 seed, u64 tick, rgb_msg *__restrict__ out,
                                                       u32 *__restrict__ scratch, unsigned char *__restrict__ stamps,
                                                       unsigned char *__restrict__ sent) {
  __shared__ u32 cnt[RGB_N_BUCKETS], rank[RGB_N_BUCKETS];
  u32 *blk = scratch + RGB_SYNTH_FIXED_WORDS + (size_t)blockIdx.x * RGB_N_BUCKETS;
  for (u32 b = threadIdx.x; b < RGB_N_BUCKETS; b += blockDim.x) { cnt[b] = 0; rank[b] = 0; }
  __syncthreads();
  const u32 G = dev.n_servers / N;
  const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
  auto bucket = [](const SynMsg &m) -> unsigned { return rgb_bucket_hinted(m.kind, m.flags, m.server, (unsigned)N, m.off_steady); };
  if (!WRITE) {
    if (g < G) synth_group<N>(dev, seed, tick, g, [&](const SynMsg &m) { atomicAdd(&cnt[bucket(m)], 1u); });
    __syncthreads();
    for (u32 b = threadIdx.x; b < RGB_N_BUCKETS; b += blockDim.x) blk[b] = cnt[b];
    return;
  }
  if (g < G)
    synth_group<N>(dev, seed, tick, g, [&](const SynMsg &m) {
      const unsigned b = bucket(m);
      const u32 slot = blk[b] + atomicAdd(&rank[b], 1u);
      syn_store(out + slot, m);
      /* the producer's own count of the messages it has addressed to the server = the value of the server's
       * sequence byte the message must find in a train launch (one lane per group, one message per server and
       * tick: nobody else touches the counter) */
      if (stamps != nullptr) {
        const u32 q = rgb_seq_index(m.server, (unsigned)N, dev.seq_stride);
        const unsigned char c = sent[q];
        sent[q] = (unsigned char)(c + 1u);
        stamps[slot] = c;
      }
    });
}

/* between the passes (one block of RGB_N_BUCKETS threads): per-block counts -> per-block bases, bucket by bucket */
__global__ void rgb_synth_scan_kernel(u32 *__restrict__ scratch, u32 n_blocks, u32 *__restrict__ kind_counts,
                                      u32 *__restrict__ d_n, u32 *__restrict__ bucket_counts) {
  __shared__ u32 tot[RGB_N_BUCKETS];
  u32 *fam_total = scratch, *bkt_total = scratch + RGB_N_FAMILIES, *bkt_base = bkt_total + RGB_N_BUCKETS;
  u32 *blk = scratch + RGB_SYNTH_FIXED_WORDS;
  const u32 b = threadIdx.x;
  u32 sum = 0;
  for (u32 k = 0; k < n_blocks; ++k) sum += blk[(size_t)k * RGB_N_BUCKETS + b];
  tot[b] = sum;
  bkt_total[b] = sum;
  if (bucket_counts != nullptr) bucket_counts[b] = sum;
  __syncthreads();
  u32 acc = 0;
  for (u32 k = 0; k < b; ++k) acc += tot[k];
  bkt_base[b] = acc;
  for (u32 k = 0; k < n_blocks; ++k) {
    const u32 c = blk[(size_t)k * RGB_N_BUCKETS + b];
    blk[(size_t)k * RGB_N_BUCKETS + b] = acc;
    acc += c;
  }
  if (b < RGB_N_FAMILIES) {      /* family b = (class b / 2, flag b & 1) */
    u32 f = 0;
    for (u32 x = 0; x < RGB_TRAIN_SHARDS; ++x) f += tot[((b >> 1) * RGB_TRAIN_SHARDS + x) * 2u + (b & 1u)];
    fam_total[b] = f;
  }
  if (b == 0) {
    u32 total = 0;
    const unsigned kind_of_rank[RGB_N_CLASSES + 1] = {
        RGB_MSG_AER, RGB_MSG_AER_REPLY, RGB_MSG_WRITTEN, RGB_MSG_APPEND, RGB_MSG_PIPELINE_RPCS,
        RGB_MSG_REQUEST_VOTE, RGB_MSG_VOTE_RESULT, RGB_MSG_AWAIT_TIMEOUT, RGB_MSG_ELECTION_TIMEOUT,
        RGB_MSG_PRE_VOTE_RPC, RGB_MSG_PRE_VOTE_RESULT, RGB_MSG_SNAPSHOT_WRITTEN, RGB_MSG_HEARTBEAT_RPC,
        RGB_MSG_HEARTBEAT_REPLY, RGB_MSG_CONSISTENT_QUERY, RGB_MSG_NOP};
    for (unsigned c = 0; c <= RGB_N_CLASSES; ++c) {
      u32 ct = 0;
      for (u32 k = 0; k < 2u * RGB_TRAIN_SHARDS; ++k) ct += tot[c * 2u * RGB_TRAIN_SHARDS + k];
      total += ct;
      if (kind_counts != nullptr && ct) kind_counts[kind_of_rank[c]] += ct;
    }
    if (d_n != nullptr) *d_n = total;
  }
}

/* ------------------------------------------------------------- support kernels -- */

__global__ void rgb_pack_kernel(rgb_dev dev, const rgb_server_state *__restrict__ in, u32 first, u32 n) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const rgb_server_state &h = in[k];
  const u32 s = first + k;
  u64 *hot = dev.hot + (size_t)s * RGB_HOT_WORDS;
  const unsigned N = dev.n_members;
  /* canonical run table: merge adjacent equal-term runs, keep the newest max_runs */
  u64 *runs = dev.runs + (size_t)s * dev.max_runs * 2;
  unsigned nr = 0;
  u64 first_index = h.first_index;
  u64 lrs = 0, lrt = 0, prs = 0, prt = 0;
  if (h.first_index <= h.last_index) {
    unsigned src_n = h.n_runs > RGB_MAX_RUNS ? RGB_MAX_RUNS : h.n_runs;
    /* count canonical runs */
    unsigned canon = 0;
    for (unsigned r = 0; r < src_n; ++r)
      if (r == 0 || h.run_term[r] != h.run_term[r - 1]) canon++;
    unsigned skip = canon > dev.max_runs ? canon - dev.max_runs : 0;
    unsigned ci = 0;
    for (unsigned r = 0; r < src_n; ++r) {
      if (!(r == 0 || h.run_term[r] != h.run_term[r - 1])) continue;
      if (ci >= skip) {
        if (nr == 0 && skip) first_index = h.run_start[r];
        runs[2 * nr] = h.run_start[r]; runs[2 * nr + 1] = h.run_term[r];
        prs = lrs; prt = lrt;
        lrs = h.run_start[r]; lrt = h.run_term[r];
        nr++;
      }
      ci++;
    }
  }
  u64 pk = 0;
  pk = pk_set(pk, PK_ROLE_SH, 3, h.role);
  pk = pk_set(pk, PK_COND_SH, 2, h.cond_reason == RGB_COND_WAL_DOWN_LEADER ? RGB_COND_WAL_DOWN : h.cond_reason);
  pk = pk_set(pk, PK_CONDTO_SH, 1, h.cond_reason == RGB_COND_WAL_DOWN_LEADER ? 1 : 0);
  pk = pk_set(pk, PK_SELF_SH, 4, h.self);
  pk = pk_set(pk, PK_VOTES_SH, 4, h.votes);
  pk = pk_set(pk, PK_NRUNS_SH, 5, nr);
  pk = pk_set(pk, PK_NONVOTER_SH, 1, h.self_nonvoter ? 1 : 0);
  pk = pk_set(pk, PK_VOTED_SH, 4, slot8to4(h.voted_for));
  pk = pk_set(pk, PK_LEADER_SH, 4, slot8to4(h.leader_id));
  pk = pk_set(pk, PK_CONDLDR_SH, 4, slot8to4(h.cond_leader));
  pk = pk_set(pk, PK_PRESENT_SH, 8, h.present_mask);
  pk = pk_set(pk, PK_VOTER_SH, 8, h.voter_mask);
  pk = pk_set(pk, PK_STATUS_SH, 8, h.status_mask);
  hot[HOT_CT] = h.current_term; hot[HOT_CI] = h.commit_index; hot[HOT_LA] = h.last_applied;
  hot[HOT_LI] = h.last_index; hot[HOT_LT] = h.last_term;
  hot[HOT_LWI] = h.last_written_index; hot[HOT_LWT] = h.last_written_term;
  hot[HOT_PK] = pk; hot[HOT_SI] = h.snapshot_index; hot[HOT_ST] = h.snapshot_term;
  hot[HOT_FIRST] = first_index; hot[HOT_LRS] = lrs; hot[HOT_LRT] = lrt;
  if (nr < 2) { prs = 0; prt = 0; }
  hot[HOT_PRS] = prs; hot[HOT_PRT] = prt;
  hot[HOT_PEND] = h.pending_first;
  {
    u64 *q = dev.qry + (size_t)s * RGB_QRY_WORDS;
    bool peer_nz = false;
    q[0] = h.query_index;
    for (unsigned i = 0; i < 8; ++i) { q[1 + i] = h.peer_query_index[i]; peer_nz = peer_nz || h.peer_query_index[i] != 0; }
    for (unsigned i = 9; i < RGB_QRY_WORDS; ++i) q[i] = 0;
    /* {snapshot_backoff,_} peers: never normal at the same time, never self, only members */
    const unsigned backoff = (unsigned)h.backoff_mask & ~(unsigned)h.status_mask & (unsigned)h.present_mask &
                             ~(1u << h.self) & 0xFFu;
    q[QRY_BACKOFF] = backoff;
    q[QRY_TOKEN] = h.pre_vote_token;
    q[QRY_MACVER] = (u64)h.machine_version | ((u64)h.effective_machine_version << 32);
    /* sparse pending: a single old range sits in the HI slot */
    const unsigned npo = h.n_pending_old > 2 ? 2u : h.n_pending_old;
    q[QRY_PEND_LO] = npo == 2 ? h.pending_old[0][0] : 1; q[QRY_PEND_LO + 1] = npo == 2 ? h.pending_old[0][1] : 0;
    q[QRY_PEND_HI] = npo ? h.pending_old[npo - 1][0] : 1; q[QRY_PEND_HI + 1] = npo ? h.pending_old[npo - 1][1] : 0;
    pk = pk_set(pk, PK_PENDX_SH, 1, npo ? 1 : 0);
    pk = pk_set(pk, PK_BACKOFF_SH, 1, backoff ? 1 : 0);
    pk = pk_set(pk, PK_QSELF_SH, 1, h.query_index != 0 ? 1 : 0);
    pk = pk_set(pk, PK_QPEER_SH, 1, peer_nz ? 1 : 0);
    hot[HOT_PK] = pk;
  }
  u64 *pr = dev.peers + (size_t)s * dev.peer_stride;
  for (unsigned i = 0; i < dev.peer_stride; ++i) pr[i] = 0;
  for (unsigned i = 0; i < N; ++i) {
    pr[PEER_MI(i, N)] = h.match_index[i]; pr[PEER_NI(i, N)] = h.next_index[i]; pr[PEER_CS(i, N)] = h.commit_index_sent[i];
  }
  u64 *cd = dev.cond + (size_t)s * 4;
  for (int i = 0; i < 4; ++i) cd[i] = h.cond_reply[i];
}

__global__ void rgb_unpack_kernel(rgb_dev dev, rgb_server_state *__restrict__ out, u32 first, u32 n) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const u32 s = first + k;
  rgb_server_state h;
  memset(&h, 0, sizeof h);
  const u64 *hot = dev.hot + (size_t)s * RGB_HOT_WORDS;
  const unsigned N = dev.n_members;
  const u64 pk = hot[HOT_PK];
  h.current_term = hot[HOT_CT]; h.commit_index = hot[HOT_CI]; h.last_applied = hot[HOT_LA];
  h.last_index = hot[HOT_LI]; h.last_term = hot[HOT_LT];
  h.last_written_index = hot[HOT_LWI]; h.last_written_term = hot[HOT_LWT];
  h.snapshot_index = hot[HOT_SI]; h.snapshot_term = hot[HOT_ST];
  h.first_index = hot[HOT_FIRST];
  const u64 *cd = dev.cond + (size_t)s * 4;
  for (int i = 0; i < 4; ++i) h.cond_reply[i] = cd[i];
  const u64 *pr = dev.peers + (size_t)s * dev.peer_stride;
  for (unsigned i = 0; i < N; ++i) {
    h.match_index[i] = pr[PEER_MI(i, N)]; h.next_index[i] = pr[PEER_NI(i, N)]; h.commit_index_sent[i] = pr[PEER_CS(i, N)];
  }
  unsigned nr = (unsigned)pk_get(pk, PK_NRUNS_SH, 5);
  const u64 *runs = dev.runs + (size_t)s * dev.max_runs * 2;
  if (!(h.first_index <= h.last_index)) { nr = 0; h.first_index = h.last_index + 1; }
  for (unsigned r = 0; r < nr && r < RGB_MAX_RUNS; ++r) { h.run_start[r] = runs[2 * r]; h.run_term[r] = runs[2 * r + 1]; }
  h.role = (uint8_t)pk_get(pk, PK_ROLE_SH, 3);
  h.cond_reason = (uint8_t)(pk_get(pk, PK_COND_SH, 2) + pk_get(pk, PK_CONDTO_SH, 1));   /* 3 + 1 = RGB_COND_WAL_DOWN_LEADER */
  h.self = (uint8_t)pk_get(pk, PK_SELF_SH, 4);
  h.n_members = (uint8_t)N;
  h.voted_for = (uint8_t)slot4to8((unsigned)pk_get(pk, PK_VOTED_SH, 4));
  h.leader_id = (uint8_t)slot4to8((unsigned)pk_get(pk, PK_LEADER_SH, 4));
  h.votes = (uint8_t)pk_get(pk, PK_VOTES_SH, 4);
  h.n_runs = (uint8_t)nr;
  h.present_mask = (uint8_t)pk_get(pk, PK_PRESENT_SH, 8);
  h.voter_mask = (uint8_t)pk_get(pk, PK_VOTER_SH, 8);
  h.status_mask = (uint8_t)pk_get(pk, PK_STATUS_SH, 8);
  h.self_nonvoter = (uint8_t)pk_get(pk, PK_NONVOTER_SH, 1);
  h.cond_leader = (uint8_t)slot4to8((unsigned)pk_get(pk, PK_CONDLDR_SH, 4));
  {
    const u64 *q = dev.qry + (size_t)s * RGB_QRY_WORDS;
    h.pre_vote_token = q[QRY_TOKEN];
    h.machine_version = (uint32_t)(q[QRY_MACVER] & 0xFFFFFFFFull);
    h.effective_machine_version = (uint32_t)(q[QRY_MACVER] >> 32);
  }
  h.pending_first = hot[HOT_PEND];
  {
    const u64 *q = dev.qry + (size_t)s * RGB_QRY_WORDS;
    h.query_index = q[0];
    for (unsigned i = 0; i < 8; ++i) h.peer_query_index[i] = q[1 + i];
    h.backoff_mask = pk_get(pk, PK_BACKOFF_SH, 1) ? (uint8_t)(q[QRY_BACKOFF] & 0xFFu) : 0;
    if (pk_get(pk, PK_PENDX_SH, 1)) {
      const bool two = q[QRY_PEND_LO] <= q[QRY_PEND_LO + 1];
      h.n_pending_old = two ? 2 : 1;
      if (two) { h.pending_old[0][0] = q[QRY_PEND_LO]; h.pending_old[0][1] = q[QRY_PEND_LO + 1]; }
      h.pending_old[two ? 1 : 0][0] = q[QRY_PEND_HI]; h.pending_old[two ? 1 : 0][1] = q[QRY_PEND_HI + 1];
    }
  }
  out[k] = h;
}

/* one thread per group: the ra_leaderboard row + key_metrics gauges of that group */
__global__ void rgb_leaderboard_kernel(rgb_dev dev, rgb_leaderboard_row *__restrict__ rows, u32 n_groups) {
  const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const unsigned N = dev.n_members;
  u32 leader = RGB_NONE, n_leaders = 0;
  u64 term = 0, lead_term = 0, ci = 0, la = 0, max_ci = 0, max_la = 0;
  for (unsigned m = 0; m < N; ++m) {
    const u64 *hot = dev.hot + ((size_t)g * N + m) * RGB_HOT_WORDS;
    const ulonglong2 h0 = reinterpret_cast<const ulonglong2 *>(hot)[HOT_P_TERM];
    const ulonglong2 h1 = reinterpret_cast<const ulonglong2 *>(hot)[HOT_P_CI];
    const u64 pk = hot[HOT_PK];
    const u64 ct = h0.x;
    if (ct > term) term = ct;
    if (h1.x > max_ci) max_ci = h1.x;
    if (h1.y > max_la) max_la = h1.y;
    if (pk_get(pk, PK_ROLE_SH, 3) == RGB_ROLE_LEADER) {
      n_leaders++;
      if (leader == RGB_NONE || ct > lead_term) { leader = m; lead_term = ct; ci = h1.x; la = h1.y; }
    }
  }
  rgb_leaderboard_row r;
  r.leader = leader; r.n_leaders = n_leaders; r.term = term;
  r.commit_index = leader == RGB_NONE ? max_ci : ci;
  r.last_applied = leader == RGB_NONE ? max_la : la;
  rows[g] = r;
}

__device__ __forceinline__ u64 fnv_word(u64 h, u64 w) {
#pragma unroll
  for (int i = 0; i < 8; ++i) { h ^= (w >> (8 * i)) & 0xFFull; h *= 0x100000001B3ull; }
  return h;
}

/* canonical-state checksum per server; same word order as oracle's ora_server_checksum */
__global__ void rgb_checksum_kernel(rgb_dev dev, u32 first, u32 n, u64 *__restrict__ out) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const u32 s = first + k;
  const u64 *hot = dev.hot + (size_t)s * RGB_HOT_WORDS;
  const unsigned N = dev.n_members;
  const u64 pk = hot[HOT_PK];
  u64 li = hot[HOT_LI], fi = hot[HOT_FIRST];
  unsigned nr = (unsigned)pk_get(pk, PK_NRUNS_SH, 5);
  if (!(fi <= li)) { nr = 0; fi = li + 1; }
  u64 x = 0xCBF29CE484222325ull;
  x = fnv_word(x, hot[HOT_CT]); x = fnv_word(x, hot[HOT_CI]); x = fnv_word(x, hot[HOT_LA]);
  x = fnv_word(x, li); x = fnv_word(x, hot[HOT_LT]); x = fnv_word(x, hot[HOT_LWI]);
  x = fnv_word(x, hot[HOT_LWT]); x = fnv_word(x, hot[HOT_SI]); x = fnv_word(x, hot[HOT_ST]);
  x = fnv_word(x, fi);
  u64 packed = pk_get(pk, PK_ROLE_SH, 3) | ((pk_get(pk, PK_COND_SH, 2) + pk_get(pk, PK_CONDTO_SH, 1)) << 8) |
               (pk_get(pk, PK_SELF_SH, 4) << 16) | ((u64)N << 24) |
               ((u64)slot4to8((unsigned)pk_get(pk, PK_VOTED_SH, 4)) << 32) |
               ((u64)slot4to8((unsigned)pk_get(pk, PK_LEADER_SH, 4)) << 40) |
               (pk_get(pk, PK_VOTES_SH, 4) << 48) | ((u64)nr << 56);
  x = fnv_word(x, packed);
  u64 masks = pk_get(pk, PK_PRESENT_SH, 8) | (pk_get(pk, PK_VOTER_SH, 8) << 8) |
              (pk_get(pk, PK_STATUS_SH, 8) << 16) | (pk_get(pk, PK_NONVOTER_SH, 1) << 24);
  if (pk_get(pk, PK_BACKOFF_SH, 1)) masks |= ((dev.qry + (size_t)s * RGB_QRY_WORDS)[QRY_BACKOFF] & 0xFFull) << 32;
  x = fnv_word(x, masks);
  x = fnv_word(x, (dev.qry + (size_t)s * RGB_QRY_WORDS)[QRY_TOKEN]);
  x = fnv_word(x, hot[HOT_PEND]);
  if (pk_get(pk, PK_PENDX_SH, 1)) {      /* the old pending ranges, ascending */
    const u64 *q = dev.qry + (size_t)s * RGB_QRY_WORDS;
    if (q[QRY_PEND_LO] <= q[QRY_PEND_LO + 1]) { x = fnv_word(x, q[QRY_PEND_LO]); x = fnv_word(x, q[QRY_PEND_LO + 1]); }
    x = fnv_word(x, q[QRY_PEND_HI]); x = fnv_word(x, q[QRY_PEND_HI + 1]);
  }
  {
    const u64 *q = dev.qry + (size_t)s * RGB_QRY_WORDS;
    x = fnv_word(x, q[0]);
    for (unsigned i = 0; i < N && i < 8; ++i) x = fnv_word(x, q[1 + i]);
  }
  x = fnv_word(x, (dev.qry + (size_t)s * RGB_QRY_WORDS)[QRY_MACVER]);
  const u64 *pr = dev.peers + (size_t)s * dev.peer_stride;
  for (unsigned i = 0; i < N; ++i) {
    x = fnv_word(x, pr[PEER_MI(i, N)]); x = fnv_word(x, pr[PEER_NI(i, N)]); x = fnv_word(x, pr[PEER_CS(i, N)]);
  }
  const u64 *runs = dev.runs + (size_t)s * dev.max_runs * 2;
  for (unsigned r = 0; r < nr; ++r) { x = fnv_word(x, runs[2 * r]); x = fnv_word(x, runs[2 * r + 1]); }
  out[k] = x;
}

/* rgb_submit's host side made lighter:
 * rgb_stamp_rounds_kernel   a train's stamps from what the device knows: in = the round of every message (the host's
 *                           sub-tick round, one byte), out = the server's sequence byte as it stands before the launch +
 *                           the round (a server's rounds in one batch are 0, 1, 2, ..: its r-th message finds exactly
 *                           that) -- no host mirror of the sequence bytes, no per-message random access on the host
 * (decisions back to submission order, rpc records compacted: rgb_results_kernel below) */
__global__ void rgb_stamp_rounds_kernel(rgb_dev dev, const rgb_msg *__restrict__ msgs, u32 n, unsigned char *__restrict__ stamps) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 w = *reinterpret_cast<const u64 *>(msgs + i);
  const u32 sv = (u32)(w & 0xFFFFFFFFull);
  if (((w >> 32) & 0xFFull) == RGB_MSG_NOP || sv >= dev.n_servers) { stamps[i] = 0; return; }
  stamps[i] = (unsigned char)(dev.seq[rgb_seq_index(sv, dev.n_members, dev.seq_stride)] + stamps[i]);
}
/* ---- what a batch of rgb_submit hands back, written by the device INTO THE PINNED SLOT (round 6) ----
 * Two kernels behind a batch's launches replace count + un-permute + three device-to-host copies:
 *   rgb_results_sums_kernel   rpc records per block of RGB_RES_BLOCK messages, in SUBMISSION order
 *   rgb_results_kernel        every block: its base = the sums in front of it; the decisions of its messages
 *                             expanded, in submission order; its rpc records COMPACTED in (message, slot) order with
 *                             msg_index = the submission index (what rgb_collect used to do record by record on the
 *                             host) -- both staged in LDS and written out as contiguous 16- / 8-byte-per-lane stores
 *                             straight into host memory (hipHostMalloc: the device writes across PCIe; visible to the
 *                             host once the slot's event has completed).  The last block leaves the header:
 *                             [0] rpc records, [1] the train launch's error word, [2] a message reported more
 *                             records than it has slots (a kind that cannot emit rpcs did).
 * Before: the decisions went device -> device (un-permute) -> host, and the rpc records as the SPAN of fixed slots
 * between the first and the last class that can emit any -- (N-1) x 56 bytes per message of nearly the whole batch,
 * 3.5 x the decisions' bytes for groups of five, almost all of it empty slots. */
#define RGB_RES_BLOCK 128u
#define RGB_RES_MAX_STRIDE 7u       /* rpc slots per message: groups of up to eight members */
__global__ __launch_bounds__(RGB_RES_BLOCK) void rgb_results_sums_kernel(const rgb_decision *__restrict__ dec, const u32 *__restrict__ pos,
                                                                         u32 n, u32 rpc_stride, u32 *__restrict__ block_sums,
                                                                         u32 *__restrict__ err) {
  __shared__ u32 part[RGB_RES_BLOCK / 64u];
  const u32 i = blockIdx.x * RGB_RES_BLOCK + threadIdx.x;
  u32 v = i < n ? (u32)((reinterpret_cast<const u64 *>(dec + pos[i])[0] >> 48) & 0xFFull) : 0u;
  if (v > rpc_stride) { atomicOr(err, 1u); v = rpc_stride; }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 t = 0;
#pragma unroll
    for (u32 w = 0; w < RGB_RES_BLOCK / 64u; ++w) t += part[w];
    block_sums[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(RGB_RES_BLOCK) void rgb_results_kernel(const ulonglong2 *__restrict__ dec, const u32 *__restrict__ pos, u32 n,
                                                                    const u32 *__restrict__ block_sums, const u64 *__restrict__ rpcs,
                                                                    u32 rpc_stride, u32 *__restrict__ err, const u32 *__restrict__ ctl,
                                                                    ulonglong2 *__restrict__ out_dec, u64 *__restrict__ out_rpcs,
                                                                    u32 *__restrict__ out_hdr) {
  __shared__ ulonglong2 sdec[RGB_RES_BLOCK * 4u];                              /* 8 KiB */
  __shared__ u64 srec[RGB_RES_BLOCK * RGB_RES_MAX_STRIDE * 7u];                /* 49 KiB */
  __shared__ u32 sscan[RGB_RES_BLOCK];
  __shared__ u32 sred[RGB_RES_BLOCK];
  const u32 tid = threadIdx.x, b = blockIdx.x;
  const u32 i = b * RGB_RES_BLOCK + tid;
  /* the records in front of this block */
  u32 partial = 0;
  for (u32 k = tid; k < b; k += RGB_RES_BLOCK) partial += block_sums[k];
  sred[tid] = partial;
  /* this lane's decision (expanded: rgb_decision_expand of include/ra_gpu_batch.h) and its record count */
  u32 nr = 0, p = 0;
  if (i < n) {
    p = pos[i];
    const ulonglong2 *src = dec + (size_t)p * 4u;
    ulonglong2 a = src[0], bb = src[1], c, e;
    nr = (u32)((a.x >> 48) & 0xFFull);
    if (nr > rpc_stride) nr = rpc_stride;
    const u32 flags = (u32)a.y;
    if (flags & RGB_F_COMPACT) {
      const u32 aux = (u32)(a.y >> 32), f = flags & ~(u32)RGB_F_COMPACT;
      const u64 A = bb.x, B = bb.y;
      u64 w2 = 0, w3 = 0, w4 = 0, w5 = 0, ci, la;
      if (f & RGB_F_REPLY) {
        w3 = A + 1ull; w2 = B; w4 = A - (u64)(aux & 0xFFu); w5 = B - (u64)((aux >> 8) & 0xFu);
        ci = A + (u64)((aux >> 12) & 0x3FFu) - 512ull; la = A + 1ull - (u64)((aux >> 22) & 0x3FFu);
      } else if (f & RGB_F_WROTE) {
        w4 = A; w3 = A - (u64)(aux & 0xFFFFu); ci = B; la = A - (u64)(aux >> 16);
      } else { ci = A; la = B; }
      a.y = (u64)f;
      bb = make_ulonglong2(w2, w3); c = make_ulonglong2(w4, w5); e = make_ulonglong2(ci, la);
    } else { c = src[2]; e = src[3]; }
    sdec[tid * 4u + 0u] = a; sdec[tid * 4u + 1u] = bb; sdec[tid * 4u + 2u] = c; sdec[tid * 4u + 3u] = e;
  }
  sscan[tid] = nr;
  __syncthreads();
  /* inclusive scan of the counts, tree sum of the partials (128 lanes: seven steps each) */
  for (u32 o = 1; o < RGB_RES_BLOCK; o <<= 1) {
    const u32 add = tid >= o ? sscan[tid - o] : 0u;
    const u32 r2 = (tid + o < RGB_RES_BLOCK && (tid & (2u * o - 1u)) == 0u) ? sred[tid + o] : 0u;
    __syncthreads();
    sscan[tid] += add;
    sred[tid] += r2;
    __syncthreads();
  }
  const u32 base = sred[0], mine = sscan[RGB_RES_BLOCK - 1u], off = sscan[tid] - nr;
  /* the records of this lane's message, slot order, msg_index := the submission index */
  for (u32 q = 0; q < nr; ++q) {
    const u64 *r = rpcs + ((size_t)p * rpc_stride + q) * 7u;
    u64 *d = srec + (size_t)(off + q) * 7u;
    d[0] = (r[0] & 0xFFFFFFFF00000000ull) | (u64)i;
#pragma unroll
    for (int k = 1; k < 7; ++k) d[k] = r[k];
  }
  __syncthreads();
  /* out: contiguous stores, lane stride 16 / 8 bytes */
  const u32 first = b * RGB_RES_BLOCK, cnt = n - first < RGB_RES_BLOCK ? n - first : RGB_RES_BLOCK;
  ulonglong2 *od = out_dec + (size_t)first * 4u;
  for (u32 k = tid; k < cnt * 4u; k += RGB_RES_BLOCK) od[k] = sdec[k];
  u64 *orp = out_rpcs + (size_t)base * 7u;
  for (u32 k = tid; k < mine * 7u; k += RGB_RES_BLOCK) orp[k] = srec[k];
  if (b == gridDim.x - 1u && tid == 0u) {
    out_hdr[0] = base + mine;
    out_hdr[1] = ctl ? ctl[0] : 0u;
    out_hdr[2] = err[0];
    err[0] = 0u;                      /* for the slot's next batch (no memset command in front of every batch) */
  }
}

/* Undo log of a batch (rgb_submit's fail-safe, rgb_api.hip): every row of the servers ids[0..n) -- hot, peers, run
 * table, cond, qry, sequence byte -- copied to undo (restore = 0) or back (restore = 1), one lane per 16-byte piece */
__host__ __device__ __forceinline__ u32 rgb_undo_pieces_of(const rgb_dev &dev) {
  return RGB_HOT_WORDS / 2u + dev.peer_stride / 2u + dev.max_runs + 2u + RGB_QRY_WORDS / 2u + 1u;
}
__global__ void rgb_undo_kernel(rgb_dev dev, const u32 *__restrict__ ids, u32 n, ulonglong2 *__restrict__ undo, u32 restore) {
  const u32 P = rgb_undo_pieces_of(dev);
  const u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u32 i = (u32)(idx / P), p = (u32)(idx - (u64)i * P);
  if (i >= n) return;
  const u32 sv = ids[i];
  if (sv >= dev.n_servers) return;
  ulonglong2 *slot = undo + (size_t)i * P + p;
  u32 q = p;
  ulonglong2 *row;
  if (q < RGB_HOT_WORDS / 2u) row = reinterpret_cast<ulonglong2 *>(dev.hot + (size_t)sv * RGB_HOT_WORDS) + q;
  else if ((q -= RGB_HOT_WORDS / 2u) < dev.peer_stride / 2u) row = reinterpret_cast<ulonglong2 *>(dev.peers + (size_t)sv * dev.peer_stride) + q;
  else if ((q -= dev.peer_stride / 2u) < dev.max_runs) row = reinterpret_cast<ulonglong2 *>(dev.runs + (size_t)sv * dev.max_runs * 2u) + q;
  else if ((q -= dev.max_runs) < 2u) row = reinterpret_cast<ulonglong2 *>(dev.cond + (size_t)sv * 4u) + q;
  else if ((q -= 2u) < RGB_QRY_WORDS / 2u) row = reinterpret_cast<ulonglong2 *>(dev.qry + (size_t)sv * RGB_QRY_WORDS) + q;
  else {
    unsigned char *b = dev.seq + rgb_seq_index(sv, dev.n_members, dev.seq_stride);
    if (restore) *b = (unsigned char)slot->x; else *slot = make_ulonglong2((u64)*b, 0);
    return;
  }
  if (restore) *row = *slot; else *slot = *row;
}

}  // namespace

#define RGB_BLOCK 256
/* variant builds of tools/build_variants.sh instantiate one group size only (-DRGB_X_ONLY_N=5): seconds instead
 * of minutes per build */
#ifdef RGB_X_ONLY_N
#define RGB_LAUNCH_ALL_N LAUNCH(RGB_X_ONLY_N)
#else
#define RGB_LAUNCH_ALL_N LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4) LAUNCH(5) LAUNCH(6) LAUNCH(7) LAUNCH(8)
#endif

template <int KIND, bool SEQXP = (KIND < 0)>
static int launch_tick_kind(const rgb_dev &dev, const rgb_msg *d_msgs, u32 n, const u32 *d_n, rgb_decision *d_dec,
                            rgb_rpc *d_rpcs, u32 rpc_slot_base, u32 msg_index_base, hipStream_t st) {
  dim3 grid((n + RGB_TICK_BLOCK - 1) / RGB_TICK_BLOCK), block(RGB_TICK_BLOCK);
#define LAUNCH(NN)                                                                                   \
  case NN:                                                                                           \
    hipLaunchKernelGGL((rgb_tick_kernel<NN, KIND, SEQXP>), grid, block, 0, st, dev, d_msgs, n, d_n, d_dec, \
                       d_rpcs, rpc_slot_base, msg_index_base);                                       \
    break;
  switch (dev.n_members) {
    RGB_LAUNCH_ALL_N
    default: return -1;
  }
#undef LAUNCH
  return (int)hipGetLastError();
}

int rgb_launch_tick(const rgb_dev &dev, int cls, const rgb_msg *d_msgs, u32 n, const u32 *d_n,
                    rgb_decision *d_dec, rgb_rpc *d_rpcs, u32 rpc_slot_base, u32 msg_index_base, void *stream) {
  (void)hipGetLastError();   /* a stale error of an earlier call in this thread is not this launch's */
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  switch (cls) {
    case 0: return launch_tick_kind<RGB_MSG_AER>(dev, d_msgs, n, d_n, d_dec, d_rpcs, rpc_slot_base, msg_index_base, st);
    case 1: return launch_tick_kind<RGB_MSG_AER_REPLY>(dev, d_msgs, n, d_n, d_dec, d_rpcs, rpc_slot_base, msg_index_base, st);
    case 2: return launch_tick_kind<RGB_MSG_WRITTEN>(dev, d_msgs, n, d_n, d_dec, d_rpcs, rpc_slot_base, msg_index_base, st);
    case 3: return launch_tick_kind<RGB_MSG_APPEND>(dev, d_msgs, n, d_n, d_dec, d_rpcs, rpc_slot_base, msg_index_base, st);
    /* written events that may carry more than two ranges (RGB_MF_SEQX: the written class of an rgb_submit_seq batch) */
    case RGB_TICK_CLS_WRITTEN_SEQX: return launch_tick_kind<RGB_MSG_WRITTEN, true>(dev, d_msgs, n, d_n, d_dec, d_rpcs, rpc_slot_base, msg_index_base, st);
    /* NOP slots only (the tail of a round of rgb_submit): their empty decisions */
    case RGB_TICK_CLS_NOP: return launch_tick_kind<RGB_MSG_NOP>(dev, d_msgs, n, d_n, d_dec, d_rpcs, rpc_slot_base, msg_index_base, st);
    default: return launch_tick_kind<-1>(dev, d_msgs, n, d_n, d_dec, d_rpcs, rpc_slot_base, msg_index_base, st);
  }
}

int rgb_launch_tick_classes(const rgb_dev &dev, const rgb_msg *d_msgs, const u32 counts[RGB_N_CLASSES],
                            const u32 *d_family_totals, u32 max_msgs, rgb_decision *d_dec, rgb_rpc *d_rpcs,
                            u32 rpc_slot_base, u32 msg_index_base, void *stream) {
  (void)hipGetLastError();   /* a stale error of an earlier call in this thread is not this launch's */
  hipStream_t st = (hipStream_t)stream;
  u32 n[RGB_N_CLASSES];
  for (int c = 0; c < RGB_N_CLASSES; ++c) n[c] = counts ? counts[c] : 0;
  rgb_tick_plan plan;
  rgb_make_plan(n, plan, dev.n_members);
  u32 blocks = plan.blk_end[RGB_N_CLASSES - 1];
  /* device-side counts: enough blocks for any split of max_msgs into classes; surplus blocks return at once */
  if (d_family_totals) blocks = (max_msgs + 31u) / 32u + RGB_N_CLASSES;   /* 32 = the smallest slice of any class */
  if (blocks == 0) return 0;
  dim3 grid(blocks), block(RGB_TICK_BLOCK);
#define LAUNCH(NN)                                                                                     \
  case NN:                                                                                             \
    hipLaunchKernelGGL(rgb_tick_classes_kernel<NN>, grid, block, 0, st, dev, d_msgs, plan, d_family_totals, \
                       d_dec, d_rpcs, rpc_slot_base, msg_index_base);                                  \
    break;
  switch (dev.n_members) {
    RGB_LAUNCH_ALL_N
    default: return -1;
  }
#undef LAUNCH
  return (int)hipGetLastError();
}

u32 rgb_synth_scratch_words(u32 n_groups) { return RGB_SYNTH_FIXED_WORDS + ((n_groups + 63u) / 64u) * RGB_N_BUCKETS; }

int rgb_launch_synth(const rgb_dev &dev, u64 seed, u64 tick, rgb_msg *d_msgs, u32 *d_scratch,
                     u32 *d_kind_counts, u32 *d_n, u32 *d_bucket_counts, unsigned char *d_stamps, unsigned char *d_sent,
                     void *stream) {
  (void)hipGetLastError();   /* a stale error of an earlier call in this thread is not this launch's */
  hipStream_t st = (hipStream_t)stream;
  const u32 G = dev.n_servers / dev.n_members;
  const u32 nblk = (G + 63) / 64;
  dim3 grid(nblk), block(64);
  hipError_t e = hipMemsetAsync(d_scratch, 0, RGB_SYNTH_FIXED_WORDS * sizeof(u32), st);
  if (e != hipSuccess) return (int)e;
#define LAUNCH(NN)                                                                                     \
  case NN:                                                                                             \
    hipLaunchKernelGGL((rgb_synth_kernel<NN, false>), grid, block, 0, st, dev, seed, tick, d_msgs, d_scratch, \
                       (unsigned char *)nullptr, (unsigned char *)nullptr);                             \
    hipLaunchKernelGGL(rgb_synth_scan_kernel, dim3(1), dim3(RGB_N_BUCKETS), 0, st, d_scratch, nblk, d_kind_counts, \
                       d_n, d_bucket_counts);                                                          \
    hipLaunchKernelGGL((rgb_synth_kernel<NN, true>), grid, block, 0, st, dev, seed, tick, d_msgs, d_scratch, \
                       d_stamps, d_sent);                                                              \
    break;
  switch (dev.n_members) {
    RGB_LAUNCH_ALL_N
    default: return -1;
  }
#undef LAUNCH
  return (int)hipGetLastError();
}

/* The plan of one train tick from its bucket counts.  A class takes as many ROWS (of RGB_TRAIN_SHARDS blocks) as its
 * fullest shard has slices; row j of class c serves slice j of every shard.  The rows of all classes are interleaved
 * by RELATIVE POSITION (j + 1/2) / rows(c): buckets are in group order, so the messages of one group range sit at the
 * same place of the block order whatever their class -- and a server's next message, whatever ITS class, comes one
 * whole tick of blocks after the previous one: the wavefront that serves it finds its dependencies committed instead of
 * holding a slot while it waits.  row_tab[k] = plan class << 24 | row of the plan class (plan class = 2 x class + sub-bucket,
 * rgb_internal.h); returns the number of rows. */
/* how much earlier than its group range's place in the tick a class's rows start, in ticks (measured wavefront
 * lives, tools/train_timeline.py, relative to the append_entries_rpc class; index = class rank) */
float rgb_train_lead[RGB_N_CLASSES] = {0.0f, 0.15f, 0.10f, 0.08f, 0.08f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.25f, 0.0f, 0.0f, 0.0f};
extern "C" void rgb_train_set_lead(const float *lead) {      /* tuning hook of tools/ (not part of the boundary) */
  for (int c = 0; c < RGB_N_CLASSES; ++c) rgb_train_lead[c] = lead[c];
}

/* the merge key of row j of a plan class with `rows` rows, as ONE expression for the host's merge and the device's
 * rank computation (rgb_train_plan_kernel): no contraction into a fused multiply-add, so that both produce the same
 * doubles and therefore the same table */
static inline __host__ __device__ double rgb_plan_key(u32 j, double step, double lead) {
#pragma clang fp contract(off)
  const double pos = (2.0 * (double)j + 1.0) * 0.5 * step;
  return pos - lead;
}

u32 rgb_train_make_tick(const u32 *bucket_counts, unsigned n_members, rgb_train_tick *out, u32 *row_tab, u32 row_cap,
                        u32 snap_rows) {
  u32 acc = 0;
  u32 rows_of[RGB_N_PCLASSES + 1u];
  rows_of[RGB_PC_SNAP] = snap_rows;
  for (unsigned pc = 0; pc < RGB_N_PCLASSES + 2u; ++pc) {      /* (the two sub-buckets of the NOP class only advance acc) */
    const unsigned c = pc >> 1, sub = pc & 1u;
    u32 need = 0;
    for (unsigned x = 0; x < RGB_TRAIN_SHARDS; ++x) {
      /* the stream is bucket-major: (class, shard, sub) -- the two sub-buckets of one shard are neighbours */
      const u32 n = bucket_counts[(c * RGB_TRAIN_SHARDS + x) * 2u + sub];
      if (pc < RGB_N_PCLASSES) {
        u32 before = 0;                                          /* messages of the class in front of (shard x, sub) */
        for (unsigned y = 0; y < RGB_TRAIN_SHARDS; ++y)
          before += (y < x ? bucket_counts[(c * RGB_TRAIN_SHARDS + y) * 2u] + bucket_counts[(c * RGB_TRAIN_SHARDS + y) * 2u + 1u] : 0u);
        if (sub) before += bucket_counts[(c * RGB_TRAIN_SHARDS + x) * 2u];
        out->off[pc][x] = acc + before; out->cnt[pc][x] = n;
        const u32 sl = rgb_train_class_slice((int)c, n_members);
        const u32 r = (n + sl - 1) / sl;
        if (r > need) need = r;
      }
    }
    if (pc < RGB_N_PCLASSES) rows_of[pc] = need;
    if (sub) {                                                   /* the class is done: its messages are behind us */
      for (unsigned x = 0; x < RGB_TRAIN_SHARDS; ++x)
        acc += bucket_counts[(c * RGB_TRAIN_SHARDS + x) * 2u] + bucket_counts[(c * RGB_TRAIN_SHARDS + x) * 2u + 1u];
    }
  }
  u32 total = 0;
  for (unsigned pc = 0; pc <= RGB_N_PCLASSES; ++pc) total += rows_of[pc];
  out->n_rows = total;
  out->msg_base = 0;
  out->snap = 0;
  if (row_tab == nullptr || total > row_cap) return total;
  /* merge by key (j + 1/2) / rows(pc) - lead(class): a class whose wavefronts live longer starts that much earlier, so
   * that what lines up from tick to tick is the time a group range's messages COMMIT, not the time they start: a
   * wavefront's dependencies then have the slack of (cadence - its own life) whatever class committed them
   * (without the leads the slowest classes -- snapshot_written, the leader-side ones -- commit later than one tick
   * after their predecessors start, the wavefronts that depend on them wait holding their slots, live longer
   * themselves, and the waits cascade).  rgb_train_lead[] is in ticks; ties: the heavier class first. */
  /* a merge of the non-empty plan classes in heaviest-first order (the tie break), keys advanced by addition: this runs
   * on the host once per tick of every plan (~4 us for the 65 536 x 5 closed loop) */
  int act[RGB_N_PCLASSES + 1]; u32 next[RGB_N_PCLASSES + 1]; double key[RGB_N_PCLASSES + 1], step[RGB_N_PCLASSES + 1];
  double lead_of[RGB_N_PCLASSES + 1];
  unsigned n_act = 0;
  if (snap_rows) {                                           /* (first: it wins every tie) */
    act[0] = (int)RGB_PC_SNAP; next[0] = 0; step[0] = 1.0 / (double)snap_rows; lead_of[0] = (double)RGB_SNAP_LEAD;
    key[0] = rgb_plan_key(0, step[0], lead_of[0]);
    n_act = 1;
  }
  for (unsigned q = 0; q < RGB_N_CLASSES; ++q) {
    const int c = rgb_class_at(q);
    for (int sub = 0; sub < 2; ++sub) {
      const int pc = 2 * c + sub;
      if (rows_of[pc] == 0) continue;
      act[n_act] = pc; next[n_act] = 0;
      step[n_act] = 1.0 / (double)rows_of[pc];
      lead_of[n_act] = (double)rgb_train_lead[c];
      key[n_act] = rgb_plan_key(0, step[n_act], lead_of[n_act]);
      n_act += 1;
    }
  }
  for (u32 k = 0; k < total; ++k) {
    unsigned best = 0;
    for (unsigned a = 1; a < n_act; ++a)
      if (key[a] < key[best]) best = a;
    row_tab[k] = ((u32)act[best] << 24) | next[best];
    next[best] += 1;
    if (next[best] >= rows_of[act[best]]) key[best] = 1e300;     /* exhausted */
    else key[best] = rgb_plan_key(next[best], step[best], lead_of[best]);
  }
  return total;
}

/* The same plan built ON THE DEVICE (round 5): one block per tick, from the bucket counts a device-side producer left
 * in device memory -- no copy of the counts to the host, no host merge, no upload of the tables between the producer
 * and the launch (rgb_train_plan_build_device).  Offsets = the exclusive prefix sum over the tick's 256 buckets; rows
 * per plan class = its fullest shard's slices; the row table by RANK: the place of row j of plan class a in the merged
 * order is the number of rows with a smaller key (or an equal key and an earlier place in the heaviest-first list),
 * counted class by class with a binary search over that class's keys -- the same keys, as the same doubles
 * (rgb_plan_key), as the host's merge compares: the two tables are equal bit for bit (tests/test_train.py). */
struct rgb_plan_leads { float lead[RGB_N_CLASSES]; };
#ifdef RGB_HOST_EMULATION
#define RGB_PLAN_THREADS 256      /* (a fiber per lane: the test build keeps the block small) */
#else
#define RGB_PLAN_THREADS 1024     /* one row per thread for ticks of up to 1024 rows */
#endif
__global__ __launch_bounds__(1024) void rgb_train_plan_kernel(const u32 *__restrict__ bucket_counts,
                                                             rgb_train_tick *__restrict__ ticks, u32 *__restrict__ rows,
                                                             u32 rpt, u32 first_tick, u32 snapshot_every, u32 snap_rows_n,
                                                             u32 n_members, rgb_plan_leads LD, u32 *__restrict__ err) {
  __shared__ u32 bc[RGB_N_BUCKETS], pre[RGB_N_BUCKETS];
  __shared__ u32 rows_of[RGB_N_PCLASSES + 1u], cum[RGB_N_PCLASSES + 2u], nb_s[RGB_N_PCLASSES + 1u], n_act_s;
  __shared__ int act[RGB_N_PCLASSES + 1u];
  __shared__ double step_s[RGB_N_PCLASSES + 1u], lead_s[RGB_N_PCLASSES + 1u];
  const u32 t = blockIdx.x, gt = first_tick + t, tid = threadIdx.x;
  const u32 *cnts = bucket_counts + (size_t)t * RGB_N_BUCKETS;
  rgb_train_tick *out = ticks + gt;
  /* exclusive prefix sum over the 256 buckets (Hillis-Steele in LDS: eight steps) */
  static_assert(RGB_N_BUCKETS == 256u, "one bucket per thread of the first 256");
  u32 mine = 0;
  if (tid < RGB_N_BUCKETS) { mine = cnts[tid]; bc[tid] = mine; pre[tid] = mine; }
  __syncthreads();
#pragma unroll 1
  for (u32 d = 1; d < RGB_N_BUCKETS; d <<= 1) {
    u32 v = 0;
    if (tid < RGB_N_BUCKETS) v = pre[tid] + (tid >= d ? pre[tid - d] : 0u);
    __syncthreads();
    if (tid < RGB_N_BUCKETS) pre[tid] = v;
    __syncthreads();
  }
  if (tid < RGB_N_BUCKETS) pre[tid] -= mine;                    /* inclusive -> exclusive */
  __syncthreads();
  const u32 snap_rows = (snapshot_every && gt && gt % snapshot_every == 0u) ? snap_rows_n : 0u;
  if (tid < RGB_N_PCLASSES) {
    const u32 pc = tid, c = pc >> 1, sub = pc & 1u;
    const u32 sl = rgb_train_class_slice((int)c, n_members);
    u32 need = 0;
    for (u32 x = 0; x < RGB_TRAIN_SHARDS; ++x) {
      const u32 b = (c * RGB_TRAIN_SHARDS + x) * 2u + sub;       /* the stream is bucket-major: (class, shard, sub) */
      const u32 n = bc[b];
      out->off[pc][x] = pre[b]; out->cnt[pc][x] = n;
      const u32 r = (n + sl - 1u) / sl;
      need = r > need ? r : need;
    }
    rows_of[pc] = need;
  }
  if (tid == RGB_N_PCLASSES) rows_of[RGB_PC_SNAP] = snap_rows;
  __syncthreads();
  /* the non-empty plan classes in the host's tie-break order -- the snapshot's rows first, then heaviest class first,
   * sub-bucket 0 before 1 -- one thread per candidate: its place in the list is the number of non-empty candidates in
   * front of it, its first row the sum of their rows */
  if (tid <= RGB_N_PCLASSES) {
    const u32 q = tid;                                          /* 0: snapshot; 1 + 2 x position + sub */
    const int c = q ? rgb_class_at((q - 1u) >> 1) : 0;
    const int pc = q ? 2 * c + (int)((q - 1u) & 1u) : (int)RGB_PC_SNAP;
    const u32 mine_rows = rows_of[pc];
    u32 pos = 0, first = 0;
    for (u32 e = 0; e < q; ++e) {
      const int ce = e ? rgb_class_at((e - 1u) >> 1) : 0;
      const u32 r = rows_of[e ? 2 * ce + (int)((e - 1u) & 1u) : (int)RGB_PC_SNAP];
      pos += r ? 1u : 0u; first += r;
    }
    if (mine_rows) {
      act[pos] = pc; nb_s[pos] = mine_rows; cum[pos] = first;
      step_s[pos] = 1.0 / (double)mine_rows;
      lead_s[pos] = q ? (double)LD.lead[c] : (double)RGB_SNAP_LEAD;
    }
    if (q == RGB_N_PCLASSES) {                                  /* the last candidate closes the list */
      const u32 n = pos + (mine_rows ? 1u : 0u), total = first + mine_rows;
      cum[n] = total;
      n_act_s = n;
      /* a tick whose rows do not fit the table (rgb_train_rows_bound is the bound of a tick whose classes are spread
       * evenly over the shards; a tick skewed so that different classes peak in different shards can need up to eight
       * times as many) becomes an EMPTY tick: no block of the launch looks at table entries that were never written
       * (the dealt form reads row < n_rows only, the persistent form's tickets run out at once), the launch fails
       * with RGB_TRAIN_ERR_PLAN and the ticks behind it give up on the error word instead of running stale rows */
      out->n_rows = total > rpt ? 0u : total; out->msg_base = 0;
      out->snap = snap_rows ? gt / snapshot_every : 0u;
      for (int k = 0; k < 13; ++k) out->pad[k] = 0;
      if (total > rpt) atomicOr(err, (u32)RGB_TRAIN_ERR_PLAN);     /* the table's rows do not hold this tick */
    }
  }
  __syncthreads();
  const u32 n_act = n_act_s, total = cum[n_act];
  if (total > rpt) return;
  u32 *tab = rows + (size_t)gt * rpt;
  for (u32 i = tid; i < total; i += blockDim.x) {
    u32 a = 0;
#pragma unroll 8
    for (u32 e = 1; e < n_act; ++e) a += i >= cum[e] ? 1u : 0u;  /* (independent reads: the list is short) */
    const u32 j = i - cum[a];
    const double k = rgb_plan_key(j, step_s[a], lead_s[a]);
    u32 rank = j;
#pragma unroll 4
    for (u32 b = 0; b < n_act; ++b) {
      if (b == a) continue;
      /* rows of b in front of (a, j): key_b < k, or key_b == k for the classes listed before a.  The keys of a class
       * are (j' + 1/2) / rows_b - lead_b: the count is about (k + lead_b) rows_b - 1/2 -- estimated in closed form,
       * then settled by the exact comparison of the keys themselves (the predicate is monotone in j') */
      const u32 nb = nb_s[b];
      const double sb = step_s[b], lb = lead_s[b];
      const double est = (k + lb) * (double)nb - 0.5;
      u32 lo = est <= 0.0 ? 0u : est >= (double)nb ? nb : (u32)est;
      for (;;) {                                                   /* lo = the first j' that is not in front */
        if (lo < nb) {
          const double kb = rgb_plan_key(lo, sb, lb);
          if (b < a ? kb <= k : kb < k) { ++lo; continue; }
        }
        if (lo > 0u) {
          const double kb = rgb_plan_key(lo - 1u, sb, lb);
          if (!(b < a ? kb <= k : kb < k)) { --lo; continue; }
        }
        break;
      }
      rank += lo;
    }
    tab[rank] = ((u32)act[a] << 24) | j;
  }
}

int rgb_launch_train_plan(const u32 *d_bucket_counts, rgb_train_tick *d_ticks, u32 *d_rows, u32 rpt, u32 first_tick,
                          u32 n_ticks, u32 snapshot_every, u32 n_groups, u32 n_members, u32 *d_err, void *stream) {
  (void)hipGetLastError();
  if (n_ticks == 0) return 0;
  rgb_plan_leads ld;
  for (int c = 0; c < RGB_N_CLASSES; ++c) ld.lead[c] = rgb_train_lead[c];
  hipLaunchKernelGGL(rgb_train_plan_kernel, dim3(n_ticks), dim3(RGB_PLAN_THREADS), 0, (hipStream_t)stream, d_bucket_counts, d_ticks, d_rows,
                     rpt, first_tick, snapshot_every, rgb_train_snap_rows(n_groups), n_members, ld, d_err);
  return (int)hipGetLastError();
}

u32 rgb_train_rows_bound(u32 n_servers, u32 n_members, bool with_snapshot) {
  /* a shard holds at most ceil(groups / 8) x members messages per tick; every non-empty plan class may end in a
   * partial slice; the smallest slice is 32 messages.  This bounds every tick whose plan classes have their fullest
   * shard in common (what hashing groups over the shards gives); the worst case over ALL ticks -- class k filling
   * shard k and nothing else -- is eight times that, and a table sized for it would make every launch of the dealt
   * form eight times as many (empty) blocks.  A tick beyond the bound is refused, not mis-run: rgb_train_plan_kernel
   * writes it as an empty tick and raises RGB_TRAIN_ERR_PLAN, rgb_train_make_tick returns its row count without
   * writing the table */
  const u32 groups = n_servers / n_members;
  const u32 per_shard = ((groups + RGB_TRAIN_SHARDS - 1u) / RGB_TRAIN_SHARDS) * n_members;
  return (per_shard + 31u) / 32u + RGB_N_PCLASSES + (with_snapshot ? rgb_train_snap_rows(groups) : 0u);
}

/* blocks a persistent train launch should have: every wavefront slot of the device (occupancy x compute units) */
u32 rgb_train_resident_blocks(unsigned n_members) {
#ifdef RGB_HOST_EMULATION
  (void)n_members;
  return RGB_TRAIN_SHARDS;
#else
  int device = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&device) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) return 0;
  hipError_t e = hipErrorInvalidValue;
#define LAUNCH(NN)                                                                                          \
  case NN:                                                                                                  \
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rgb_train_kernel<NN>, RGB_TICK_BLOCK, 0);     \
    break;
  switch (n_members) {
    RGB_LAUNCH_ALL_N
    default: return 0;
  }
#undef LAUNCH
  if (e != hipSuccess || per_cu <= 0) return 0;
#ifdef RGB_X_TRAIN_GRID_PCT      /* EXPERIMENT: a share of the device's wavefront slots */
  return (u32)((u64)per_cu * (u32)cus * RGB_X_TRAIN_GRID_PCT / 100u);
#endif
  return (u32)per_cu * (u32)cus;
#endif
}

int rgb_launch_train(const rgb_dev &dev, const rgb_msg *d_msgs, const unsigned char *d_stamps, u32 tick_stride,
                     const rgb_train_tick *d_plan, const u32 *d_row_tab, u32 n_ticks, u32 bpt, rgb_decision *d_dec,
                     rgb_rpc *d_rpcs, u32 rpc_ring, u32 index_base, u32 *d_ctl, u32 n_xcc, u32 n_blocks, void *stream,
                     const unsigned char *d_snap_stamps, rgb_leaderboard_row *d_snap_rows, u32 tab_rpt) {
  (void)hipGetLastError();   /* a stale error of an earlier call in this thread is not this launch's */
  /* n_blocks = 0: the DEALT form (one block per row; the caller's calibration showed round-robin dispatch) */
  const bool dealt = n_blocks == 0;
  if (dealt) { n_blocks = RGB_TRAIN_SHARDS; n_xcc = RGB_TRAIN_SHARDS; }
  if (n_ticks == 0 || bpt == 0) return 0;
  if (bpt % RGB_TRAIN_SHARDS || n_ticks > RGB_TRAIN_MAX_TICKS) return -1;
  if (n_xcc == 0 || n_xcc > RGB_TRAIN_SHARDS || (n_xcc & (n_xcc - 1u)) != 0 || n_blocks < RGB_TRAIN_SHARDS) return -1;
  hipStream_t st = (hipStream_t)stream;
#ifdef RGB_X_PROLOG_FRONT      /* A/B timing only: rounds 3-4 */
  hipLaunchKernelGGL(rgb_train_prolog_kernel, dim3(1), dim3(256), 0, st, d_ctl, 1u);
#endif
  /* never more blocks than rows: a block without a row only costs its ticket */
  const uint64_t rows = (uint64_t)n_ticks * bpt;
  dim3 grid((u32)(rows < n_blocks ? rows : n_blocks)), block(RGB_TICK_BLOCK);
  if (grid.x < RGB_TRAIN_SHARDS) grid.x = RGB_TRAIN_SHARDS;
  if (dealt) {
    if (rows > 0x7FFFFFFFull) return -1;
    grid.x = (u32)rows;      /* one block per row of the longest tick, every tick; surplus blocks exit at once */
  }
  rgb_train_args args;
  args.dev = dev; args.msgs = d_msgs; args.stamps = d_stamps; args.plan = d_plan; args.row_tab = d_row_tab;
  args.dec = d_dec; args.rpcs = d_rpcs; args.ctl = d_ctl; args.tick_stride = tick_stride;
  args.rpt = bpt / RGB_TRAIN_SHARDS; args.n_ticks = n_ticks; args.rpc_ring = rpc_ring ? rpc_ring : 1u;
  args.tab_rpt = tab_rpt ? tab_rpt : args.rpt;
  if (args.tab_rpt < args.rpt) return -1;
  args.index_base = index_base; args.n_xcc = n_xcc;
  args.snap_stamps = d_snap_stamps; args.snap_rows = d_snap_rows;
#define LAUNCH(NN)                                                                                      \
  case NN:                                                                                              \
    if (dealt) hipLaunchKernelGGL(rgb_train_dealt_kernel<NN>, grid, block, 0, st, args);                \
    else hipLaunchKernelGGL(rgb_train_kernel<NN>, grid, block, 0, st, args);                            \
    break;
  switch (dev.n_members) {
    RGB_LAUNCH_ALL_N
    default: return -1;
  }
#undef LAUNCH
  /* BEHIND the launch (round 5; rounds 3-4 ran it in front of the NEXT one, where its launch and its 5 us stood
   * between the caller's submission and the first tick): this launch's rotation marks are verified -- the error word
   * (d_ctl[0], sticky until it is read) is final when the stream reaches whatever follows -- and every per-launch word
   * (arrival and ticket counters, the marks) is cleared for the next launch on these control words, which a context
   * hands out zeroed */
#ifndef RGB_X_PROLOG_FRONT
  hipLaunchKernelGGL(rgb_train_prolog_kernel, dim3(1), dim3(256), 0, st, d_ctl, 1u);
#endif
  return (int)hipGetLastError();
}

int rgb_launch_train_verify(u32 *d_ctl, void *stream) {
  (void)hipGetLastError();   /* a stale error of an earlier call in this thread is not this launch's */
  hipLaunchKernelGGL(rgb_train_prolog_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, d_ctl, 0u);
  return (int)hipGetLastError();
}

int rgb_launch_train_seq(const rgb_dev &dev, const rgb_msg *d_msgs, u32 n, unsigned char *d_seq_cnt,
                         unsigned char *d_stamps, void *stream) {
  (void)hipGetLastError();   /* a stale error of an earlier call in this thread is not this launch's */
  hipStream_t st = (hipStream_t)stream;
  if (n) hipLaunchKernelGGL(rgb_train_seq_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dev, d_msgs, n, d_seq_cnt, d_stamps);
  return (int)hipGetLastError();
}

__global__ void rgb_seq_bump_kernel(unsigned char *__restrict__ seq, unsigned char *__restrict__ out, u32 n) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned char c = seq[i];
  if (out != nullptr) out[i] = c;
  seq[i] = (unsigned char)(c + 1u);
}

int rgb_launch_seq_bump(unsigned char *d_seq, unsigned char *d_out, u32 n_bytes, void *stream) {
  (void)hipGetLastError();   /* a stale error of an earlier call in this thread is not this launch's */
  if (n_bytes) hipLaunchKernelGGL(rgb_seq_bump_kernel, dim3((n_bytes + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, d_seq, d_out, n_bytes);
  return (int)hipGetLastError();
}

int rgb_launch_train_calibrate(u32 *d_out, void *stream) {
  (void)hipGetLastError();   /* a stale error of an earlier call in this thread is not this launch's */
  hipLaunchKernelGGL(rgb_train_calibrate_kernel, dim3(4096), dim3(64), 0, (hipStream_t)stream, d_out);
  return (int)hipGetLastError();
}

int rgb_launch_pack(const rgb_dev &dev, const rgb_server_state *d_in, u32 first, u32 n, void *stream) {
  (void)hipGetLastError();   /* a stale error of an earlier call in this thread is not this launch's */
  if (n == 0) return 0;
  hipLaunchKernelGGL(rgb_pack_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, dev, d_in, first, n);
  return (int)hipGetLastError();
}

int rgb_launch_unpack(const rgb_dev &dev, rgb_server_state *d_out, u32 first, u32 n, void *stream) {
  (void)hipGetLastError();   /* a stale error of an earlier call in this thread is not this launch's */
  if (n == 0) return 0;
  hipLaunchKernelGGL(rgb_unpack_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, dev, d_out, first, n);
  return (int)hipGetLastError();
}

u32 rgb_results_blocks(u32 n) { return (n + RGB_RES_BLOCK - 1u) / RGB_RES_BLOCK; }

/* d_scratch: rgb_results_blocks(cap) + 1 words (the block sums, then the error word at the END: zero when allocated,
 * cleared again by every launch; cap = the slot's capacity, n <= cap); out_*: the slot's PINNED
 * host buffers (or device memory: the kernel does not care); d_ctl: the train launch's error word, or NULL */
int rgb_launch_results(const rgb_decision *d_dec, const u32 *d_pos, u32 n, u32 cap, const rgb_rpc *d_rpcs, u32 rpc_stride, u32 *d_scratch,
                       const u32 *d_ctl, rgb_decision *out_dec, rgb_rpc *out_rpcs, u32 *out_hdr, void *stream) {
  (void)hipGetLastError();   /* a stale error of an earlier call in this thread is not this launch's */
  if (n == 0 || rpc_stride > RGB_RES_MAX_STRIDE) return n == 0 ? 0 : 1;   /* (hipErrorInvalidValue) */
  const u32 nb = rgb_results_blocks(n);
  u32 *err = d_scratch + rgb_results_blocks(cap);       /* (zero at allocation; the results kernel leaves it zero) */
  hipLaunchKernelGGL(rgb_results_sums_kernel, dim3(nb), dim3(RGB_RES_BLOCK), 0, (hipStream_t)stream, d_dec, d_pos, n, rpc_stride, d_scratch, err);
  hipLaunchKernelGGL(rgb_results_kernel, dim3(nb), dim3(RGB_RES_BLOCK), 0, (hipStream_t)stream, reinterpret_cast<const ulonglong2 *>(d_dec),
                     d_pos, n, (const u32 *)d_scratch, reinterpret_cast<const u64 *>(d_rpcs), rpc_stride, err, d_ctl,
                     reinterpret_cast<ulonglong2 *>(out_dec), reinterpret_cast<u64 *>(out_rpcs), out_hdr);
  return (int)hipGetLastError();
}

int rgb_launch_leaderboard(const rgb_dev &dev, rgb_leaderboard_row *d_rows, void *stream) {
  (void)hipGetLastError();   /* a stale error of an earlier call in this thread is not this launch's */
  u32 g = dev.n_servers / dev.n_members;
  if (g == 0) return 0;
  hipLaunchKernelGGL(rgb_leaderboard_kernel, dim3((g + 255) / 256), dim3(256), 0, (hipStream_t)stream, dev, d_rows, g);
  return (int)hipGetLastError();
}

int rgb_launch_checksum(const rgb_dev &dev, u32 first, u32 n, u64 *d_out, void *stream) {
  (void)hipGetLastError();   /* a stale error of an earlier call in this thread is not this launch's */
  if (n == 0) return 0;
  hipLaunchKernelGGL(rgb_checksum_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dev, first, n, d_out);
  return (int)hipGetLastError();
}

u32 rgb_undo_pieces(const rgb_dev &dev) { return rgb_undo_pieces_of(dev); }

int rgb_launch_undo(const rgb_dev &dev, const u32 *d_ids, u32 n, void *d_undo, u32 restore, void *stream) {
  (void)hipGetLastError();   /* a stale error of an earlier call in this thread is not this launch's */
  if (n == 0) return 0;
  const u64 lanes = (u64)n * rgb_undo_pieces_of(dev);
  hipLaunchKernelGGL(rgb_undo_kernel, dim3((u32)((lanes + 255u) / 256u)), dim3(256), 0, (hipStream_t)stream, dev, d_ids, n,
                     (ulonglong2 *)d_undo, restore);
  return (int)hipGetLastError();
}

int rgb_launch_stamp_rounds(const rgb_dev &dev, const rgb_msg *d_msgs, u32 n, unsigned char *d_stamps, void *stream) {
  (void)hipGetLastError();   /* a stale error of an earlier call in this thread is not this launch's */
  if (n) hipLaunchKernelGGL(rgb_stamp_rounds_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dev, d_msgs, n, d_stamps);
  return (int)hipGetLastError();
}

