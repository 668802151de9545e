/*
 * kernel_dispatch_on_cpu.cpp -- TEST INFRASTRUCTURE, split emulation build (tests/conftest.py).
 * kernel_on_cpu.cpp is compiled once per group size (-DRGB_EMU_ONLY_N=1..8: eight small units in parallel instead of
 * one three-minute unit, every external name suffixed __N<n> by emu_rename.h).  This unit owns the plain names: the
 * lane / block emulation (emu_runtime.inc, one definition per library) and one forwarder per launcher and per emu_*
 * entry point, chosen by n_members.  Launchers that do not depend on the group size go to the unit of N = 1.
 */
#define RGB_HOST_EMULATION 1
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include "emu_runtime.inc"
#include "../../ra_amd/csrc/rgb_internal.h"

#define FOR_N(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)

/* RET name(PARAMS) forwards ARGS to name__N<n>, n = NEXPR */
#define DISPATCH(RET, NAME, PARAMS, ARGS, NEXPR, FAIL)            \
  FOR_N(DECL_##NAME)                                              \
  RET NAME PARAMS {                                               \
    switch (NEXPR) { FOR_N(CASE_##NAME) default: return FAIL; }   \
  }
#define DECLN(RET, NAME, n, PARAMS) RET NAME##__N##n PARAMS;
#define CASEN(NAME, n, ARGS) case n: return NAME##__N##n ARGS;

/* ---- launchers of rgb_internal.h ---- */
#define P_TICK (const rgb_dev &dev, int cls, const rgb_msg *d_msgs, u32 n, const u32 *d_n, rgb_decision *d_dec, rgb_rpc *d_rpcs, u32 rpc_slot_base, u32 msg_index_base, void *stream)
#define A_TICK (dev, cls, d_msgs, n, d_n, d_dec, d_rpcs, rpc_slot_base, msg_index_base, stream)
#define DECL_rgb_launch_tick(n) DECLN(int, rgb_launch_tick, n, P_TICK)
#define CASE_rgb_launch_tick(n) CASEN(rgb_launch_tick, n, A_TICK)
DISPATCH(int, rgb_launch_tick, P_TICK, A_TICK, dev.n_members, -1)

#define P_CLS (const rgb_dev &dev, const rgb_msg *d_msgs, const u32 counts[RGB_N_CLASSES], const u32 *d_family_totals, u32 max_msgs, rgb_decision *d_dec, rgb_rpc *d_rpcs, u32 rpc_slot_base, u32 msg_index_base, void *stream)
#define A_CLS (dev, d_msgs, counts, d_family_totals, max_msgs, d_dec, d_rpcs, rpc_slot_base, msg_index_base, stream)
#define DECL_rgb_launch_tick_classes(n) DECLN(int, rgb_launch_tick_classes, n, P_CLS)
#define CASE_rgb_launch_tick_classes(n) CASEN(rgb_launch_tick_classes, n, A_CLS)
DISPATCH(int, rgb_launch_tick_classes, P_CLS, A_CLS, dev.n_members, -1)

#define P_SYN (const rgb_dev &dev, u64 seed, u64 tick, rgb_msg *d_msgs, u32 *d_scratch, u32 *d_kind_counts, u32 *d_n, u32 *d_bucket_counts, unsigned char *d_stamps, unsigned char *d_sent, void *stream)
#define A_SYN (dev, seed, tick, d_msgs, d_scratch, d_kind_counts, d_n, d_bucket_counts, d_stamps, d_sent, stream)
#define DECL_rgb_launch_synth(n) DECLN(int, rgb_launch_synth, n, P_SYN)
#define CASE_rgb_launch_synth(n) CASEN(rgb_launch_synth, n, A_SYN)
DISPATCH(int, rgb_launch_synth, P_SYN, A_SYN, dev.n_members, -1)

/* (the default arguments of the declaration in rgb_internal.h belong to the plain name only) */
#define P_TRAIN (const rgb_dev &dev, const rgb_msg *d_msgs, const unsigned char *d_stamps, u32 tick_stride, const rgb_train_tick *d_plan, const u32 *d_row_tab, u32 n_ticks, u32 bpt, rgb_decision *d_dec, rgb_rpc *d_rpcs, u32 rpc_ring, u32 index_base, u32 *d_ctl, u32 n_xcc, u32 n_blocks, void *stream, const unsigned char *d_snap_stamps, rgb_leaderboard_row *d_snap_rows, u32 tab_rpt)
#define A_TRAIN (dev, d_msgs, d_stamps, tick_stride, d_plan, d_row_tab, n_ticks, bpt, d_dec, d_rpcs, rpc_ring, index_base, d_ctl, n_xcc, n_blocks, stream, d_snap_stamps, d_snap_rows, tab_rpt)
#define DECL_rgb_launch_train(n) DECLN(int, rgb_launch_train, n, P_TRAIN)
#define CASE_rgb_launch_train(n) CASEN(rgb_launch_train, n, A_TRAIN)
DISPATCH(int, rgb_launch_train, P_TRAIN, A_TRAIN, dev.n_members, -1)

#define P_TSEQ (const rgb_dev &dev, const rgb_msg *d_msgs, u32 n, unsigned char *d_seq_cnt, unsigned char *d_stamps, void *stream)
#define A_TSEQ (dev, d_msgs, n, d_seq_cnt, d_stamps, stream)
#define DECL_rgb_launch_train_seq(n) DECLN(int, rgb_launch_train_seq, n, P_TSEQ)
#define CASE_rgb_launch_train_seq(n) CASEN(rgb_launch_train_seq, n, A_TSEQ)
DISPATCH(int, rgb_launch_train_seq, P_TSEQ, A_TSEQ, dev.n_members, -1)

#define P_PACK (const rgb_dev &dev, const rgb_server_state *d_in, u32 first, u32 n, void *stream)
#define A_PACK (dev, d_in, first, n, stream)
#define DECL_rgb_launch_pack(n) DECLN(int, rgb_launch_pack, n, P_PACK)
#define CASE_rgb_launch_pack(n) CASEN(rgb_launch_pack, n, A_PACK)
DISPATCH(int, rgb_launch_pack, P_PACK, A_PACK, dev.n_members, -1)

#define P_UNPACK (const rgb_dev &dev, rgb_server_state *d_out, u32 first, u32 n, void *stream)
#define A_UNPACK (dev, d_out, first, n, stream)
#define DECL_rgb_launch_unpack(n) DECLN(int, rgb_launch_unpack, n, P_UNPACK)
#define CASE_rgb_launch_unpack(n) CASEN(rgb_launch_unpack, n, A_UNPACK)
DISPATCH(int, rgb_launch_unpack, P_UNPACK, A_UNPACK, dev.n_members, -1)

#define P_LB (const rgb_dev &dev, rgb_leaderboard_row *d_rows, void *stream)
#define A_LB (dev, d_rows, stream)
#define DECL_rgb_launch_leaderboard(n) DECLN(int, rgb_launch_leaderboard, n, P_LB)
#define CASE_rgb_launch_leaderboard(n) CASEN(rgb_launch_leaderboard, n, A_LB)
DISPATCH(int, rgb_launch_leaderboard, P_LB, A_LB, dev.n_members, -1)

#define P_CK (const rgb_dev &dev, u32 first, u32 n, u64 *d_out, void *stream)
#define A_CK (dev, first, n, d_out, stream)
#define DECL_rgb_launch_checksum(n) DECLN(int, rgb_launch_checksum, n, P_CK)
#define CASE_rgb_launch_checksum(n) CASEN(rgb_launch_checksum, n, A_CK)
DISPATCH(int, rgb_launch_checksum, P_CK, A_CK, dev.n_members, -1)

#define P_STAMP (const rgb_dev &dev, const rgb_msg *d_msgs, u32 n, unsigned char *d_stamps, void *stream)
#define A_STAMP (dev, d_msgs, n, d_stamps, stream)
#define DECL_rgb_launch_stamp_rounds(n) DECLN(int, rgb_launch_stamp_rounds, n, P_STAMP)
#define CASE_rgb_launch_stamp_rounds(n) CASEN(rgb_launch_stamp_rounds, n, A_STAMP)
DISPATCH(int, rgb_launch_stamp_rounds, P_STAMP, A_STAMP, dev.n_members, -1)

#define P_UNDO (const rgb_dev &dev, const u32 *d_ids, u32 n, void *d_undo, u32 restore, void *stream)
#define A_UNDO (dev, d_ids, n, d_undo, restore, stream)
#define DECL_rgb_launch_undo(n) DECLN(int, rgb_launch_undo, n, P_UNDO)
#define CASE_rgb_launch_undo(n) CASEN(rgb_launch_undo, n, A_UNDO)
DISPATCH(int, rgb_launch_undo, P_UNDO, A_UNDO, dev.n_members, -1)

#define P_UP (const rgb_dev &dev)
#define A_UP (dev)
#define DECL_rgb_undo_pieces(n) DECLN(u32, rgb_undo_pieces, n, P_UP)
#define CASE_rgb_undo_pieces(n) CASEN(rgb_undo_pieces, n, A_UP)
DISPATCH(u32, rgb_undo_pieces, P_UP, A_UP, dev.n_members, 0u)

#define P_RB (unsigned n_members)
#define A_RB (n_members)
#define DECL_rgb_train_resident_blocks(n) DECLN(u32, rgb_train_resident_blocks, n, P_RB)
#define CASE_rgb_train_resident_blocks(n) CASEN(rgb_train_resident_blocks, n, A_RB)
DISPATCH(u32, rgb_train_resident_blocks, P_RB, A_RB, n_members, 0u)

/* group-size independent: the unit of N = 1 */
u32 rgb_synth_scratch_words__N1(u32 n_groups);
u32 rgb_synth_scratch_words(u32 n_groups) { return rgb_synth_scratch_words__N1(n_groups); }
int rgb_launch_train_verify__N1(u32 *d_ctl, void *stream);
int rgb_launch_train_verify(u32 *d_ctl, void *stream) { return rgb_launch_train_verify__N1(d_ctl, stream); }
int rgb_launch_train_calibrate__N1(u32 *d_out, void *stream);
int rgb_launch_train_calibrate(u32 *d_out, void *stream) { return rgb_launch_train_calibrate__N1(d_out, stream); }
int rgb_launch_seq_bump__N1(unsigned char *d_seq, unsigned char *d_out, u32 n_bytes, void *stream);
int rgb_launch_seq_bump(unsigned char *d_seq, unsigned char *d_out, u32 n_bytes, void *stream) { return rgb_launch_seq_bump__N1(d_seq, d_out, n_bytes, stream); }
u32 rgb_results_blocks__N1(u32 n);
u32 rgb_results_blocks(u32 n) { return rgb_results_blocks__N1(n); }
int rgb_launch_results__N1(const rgb_decision *d_dec, const u32 *d_pos, u32 n, u32 cap, const rgb_rpc *d_rpcs, u32 rpc_stride, u32 *d_scratch, const u32 *d_ctl, rgb_decision *out_dec, rgb_rpc *out_rpcs, u32 *out_hdr, void *stream);
int rgb_launch_results(const rgb_decision *d_dec, const u32 *d_pos, u32 n, u32 cap, const rgb_rpc *d_rpcs, u32 rpc_stride, u32 *d_scratch, const u32 *d_ctl, rgb_decision *out_dec, rgb_rpc *out_rpcs, u32 *out_hdr, void *stream) {
  return rgb_launch_results__N1(d_dec, d_pos, n, cap, d_rpcs, rpc_stride, d_scratch, d_ctl, out_dec, out_rpcs, out_hdr, stream);
}
int rgb_launch_train_plan__N1(const u32 *d_bucket_counts, rgb_train_tick *d_ticks, u32 *d_rows, u32 rpt, u32 first_tick, u32 n_ticks, u32 snapshot_every, u32 n_groups, u32 n_members, u32 *d_err, void *stream);
int rgb_launch_train_plan(const u32 *d_bucket_counts, rgb_train_tick *d_ticks, u32 *d_rows, u32 rpt, u32 first_tick, u32 n_ticks, u32 snapshot_every, u32 n_groups, u32 n_members, u32 *d_err, void *stream) {
  return rgb_launch_train_plan__N1(d_bucket_counts, d_ticks, d_rows, rpt, first_tick, n_ticks, snapshot_every, n_groups, n_members, d_err, stream);
}
u32 rgb_train_rows_bound__N1(u32 n_servers, u32 n_members, bool with_snapshot);
u32 rgb_train_rows_bound(u32 n_servers, u32 n_members, bool with_snapshot) { return rgb_train_rows_bound__N1(n_servers, n_members, with_snapshot); }
/* the row plan reads rgb_train_lead[] of ITS unit: the tuning hook sets every copy, the plan comes from N = 1's */
u32 rgb_train_make_tick__N1(const u32 *bucket_counts, unsigned n_members, rgb_train_tick *out, u32 *row_tab, u32 row_cap, u32 snap_rows);
u32 rgb_train_make_tick(const u32 *bucket_counts, unsigned n_members, rgb_train_tick *out, u32 *row_tab, u32 row_cap, u32 snap_rows) {
  return rgb_train_make_tick__N1(bucket_counts, n_members, out, row_tab, row_cap, snap_rows);
}
#define DECL_LEAD(n) extern "C" void rgb_train_set_lead__N##n(const float *lead);
FOR_N(DECL_LEAD)
extern "C" void rgb_train_set_lead(const float *lead) {
#define SET_LEAD(n) rgb_train_set_lead__N##n(lead);
  FOR_N(SET_LEAD)
}

/* ---- the emu_* entry points of kernel_on_cpu.cpp (its Emu begins with the rgb_dev) ---- */
static inline u32 n_of(void *h) { return ((const rgb_dev *)h)->n_members; }
extern "C" {
#define P_NEW (uint32_t n_groups, uint32_t n_members, uint32_t max_runs, uint32_t max_pipeline_count, uint32_t max_aer_batch)
#define A_NEW (n_groups, n_members, max_runs, max_pipeline_count, max_aer_batch)
#define DECL_emu_new(n) DECLN(void *, emu_new, n, P_NEW)
#define CASE_emu_new(n) CASEN(emu_new, n, A_NEW)
DISPATCH(void *, emu_new, P_NEW, A_NEW, n_members, nullptr)

#define DECL_emu_free(n) void emu_free__N##n(void *h);
#define CASE_emu_free(n) case n: emu_free__N##n(h); return;
FOR_N(DECL_emu_free)
void emu_free(void *h) { switch (n_of(h)) { FOR_N(CASE_emu_free) default: return; } }

#define DECL_emu_set_state(n) void emu_set_state__N##n(void *h, uint32_t first, uint32_t cnt, const rgb_server_state *in);
#define CASE_emu_set_state(n) case n: emu_set_state__N##n(h, first, cnt, in); return;
FOR_N(DECL_emu_set_state)
void emu_set_state(void *h, uint32_t first, uint32_t cnt, const rgb_server_state *in) { switch (n_of(h)) { FOR_N(CASE_emu_set_state) default: return; } }

#define DECL_emu_get_state(n) void emu_get_state__N##n(void *h, uint32_t first, uint32_t cnt, rgb_server_state *out);
#define CASE_emu_get_state(n) case n: emu_get_state__N##n(h, first, cnt, out); return;
FOR_N(DECL_emu_get_state)
void emu_get_state(void *h, uint32_t first, uint32_t cnt, rgb_server_state *out) { switch (n_of(h)) { FOR_N(CASE_emu_get_state) default: return; } }

#define P_STEP (void *h, const rgb_msg *msgs, uint32_t n, rgb_decision *dec, rgb_rpc *rpcs, uint32_t rpc_cap, uint32_t *n_rpcs_out, int specialised)
#define A_STEP (h, msgs, n, dec, rpcs, rpc_cap, n_rpcs_out, specialised)
#define DECL_emu_step(n) DECLN(int, emu_step, n, P_STEP)
#define CASE_emu_step(n) CASEN(emu_step, n, A_STEP)
DISPATCH(int, emu_step, P_STEP, A_STEP, n_of(h), -1)

#define P_ETICK (void *h, int cls, const rgb_msg *msgs, uint32_t n, rgb_decision *dec, rgb_rpc *rpcs, uint32_t rpc_cap, uint32_t *n_rpcs_out)
#define A_ETICK (h, cls, msgs, n, dec, rpcs, rpc_cap, n_rpcs_out)
#define DECL_emu_launch_tick(n) DECLN(int, emu_launch_tick, n, P_ETICK)
#define CASE_emu_launch_tick(n) CASEN(emu_launch_tick, n, A_ETICK)
DISPATCH(int, emu_launch_tick, P_ETICK, A_ETICK, n_of(h), -1)

#define P_ECLS (void *h, const rgb_msg *msgs, const uint32_t *counts, uint32_t n, rgb_decision *dec, rgb_rpc *rpcs, uint32_t rpc_cap, uint32_t *n_rpcs_out)
#define A_ECLS (h, msgs, counts, n, dec, rpcs, rpc_cap, n_rpcs_out)
#define DECL_emu_launch_classes(n) DECLN(int, emu_launch_classes, n, P_ECLS)
#define CASE_emu_launch_classes(n) CASEN(emu_launch_classes, n, A_ECLS)
DISPATCH(int, emu_launch_classes, P_ECLS, A_ECLS, n_of(h), -1)

#define P_ECLSD (void *h, const rgb_msg *msgs, const uint32_t *family_totals, uint32_t max_msgs, rgb_decision *dec)
#define A_ECLSD (h, msgs, family_totals, max_msgs, dec)
#define DECL_emu_launch_classes_dev(n) DECLN(int, emu_launch_classes_dev, n, P_ECLSD)
#define CASE_emu_launch_classes_dev(n) CASEN(emu_launch_classes_dev, n, A_ECLSD)
DISPATCH(int, emu_launch_classes_dev, P_ECLSD, A_ECLSD, n_of(h), -1)

#define P_EPACK (void *h, uint32_t first, uint32_t n, const rgb_server_state *in)
#define A_EPACK (h, first, n, in)
#define DECL_emu_launch_pack(n) DECLN(int, emu_launch_pack, n, P_EPACK)
#define CASE_emu_launch_pack(n) CASEN(emu_launch_pack, n, A_EPACK)
DISPATCH(int, emu_launch_pack, P_EPACK, A_EPACK, n_of(h), -1)

#define P_EUNPACK (void *h, uint32_t first, uint32_t n, rgb_server_state *out)
#define A_EUNPACK (h, first, n, out)
#define DECL_emu_launch_unpack(n) DECLN(int, emu_launch_unpack, n, P_EUNPACK)
#define CASE_emu_launch_unpack(n) CASEN(emu_launch_unpack, n, A_EUNPACK)
DISPATCH(int, emu_launch_unpack, P_EUNPACK, A_EUNPACK, n_of(h), -1)

#define P_ECK (void *h, uint32_t first, uint32_t n, uint64_t *out)
#define A_ECK (h, first, n, out)
#define DECL_emu_launch_checksum(n) DECLN(int, emu_launch_checksum, n, P_ECK)
#define CASE_emu_launch_checksum(n) CASEN(emu_launch_checksum, n, A_ECK)
DISPATCH(int, emu_launch_checksum, P_ECK, A_ECK, n_of(h), -1)

#define P_ELB (void *h, rgb_leaderboard_row *rows)
#define A_ELB (h, rows)
#define DECL_emu_launch_leaderboard(n) DECLN(int, emu_launch_leaderboard, n, P_ELB)
#define CASE_emu_launch_leaderboard(n) CASEN(emu_launch_leaderboard, n, A_ELB)
DISPATCH(int, emu_launch_leaderboard, P_ELB, A_ELB, n_of(h), -1)

#define P_ESW (void *h)
#define A_ESW (h)
#define DECL_emu_synth_scratch_words(n) DECLN(uint32_t, emu_synth_scratch_words, n, P_ESW)
#define CASE_emu_synth_scratch_words(n) CASEN(emu_synth_scratch_words, n, A_ESW)
DISPATCH(uint32_t, emu_synth_scratch_words, P_ESW, A_ESW, n_of(h), 0u)

#define P_ESYN (void *h, uint64_t seed, uint64_t tick, rgb_msg *msgs, uint32_t *scratch, uint32_t *kind_counts, uint32_t *n_out, uint32_t *bucket_counts)
#define A_ESYN (h, seed, tick, msgs, scratch, kind_counts, n_out, bucket_counts)
#define DECL_emu_launch_synth(n) DECLN(int, emu_launch_synth, n, P_ESYN)
#define CASE_emu_launch_synth(n) CASEN(emu_launch_synth, n, A_ESYN)
DISPATCH(int, emu_launch_synth, P_ESYN, A_ESYN, n_of(h), -1)
}  // extern "C"
