"""The C ABI end to end without a GPU: ra_amd/csrc/rgb_api.hip (contexts, the staging ring, rgb_submit's
sub-tick rounds and family ordering, rgb_collect's un-permutation, state upload/download, snapshots, checksums)
together with the kernels, all compiled as x86 C++ on the block emulation (tests/native).  ra_amd.engine is
bound to that library in this test process only, and the C-ABI parity tests of tests/test_gpu_parity.py and
tests/test_cluster_safety.py are re-run through it unchanged.  (Tests that need torch device tensors -- the
device-resident paths -- only run on the GPU.)"""
import numpy as np
import pytest

import test_gpu_parity as G
import vector_runner as VR
from ra_amd import abi


def test_all_reference_vectors_through_the_c_abi(emulated_engine):
    for v in G.DATA["vectors"]:
        try:
            G.test_hip_matches_reference_vector(emulated_engine, v)
        except AssertionError as e:
            raise AssertionError(f"vector {v['id']}: {e}") from e


@pytest.mark.parametrize("n_members,seed,groups", [(3, 101, 300), (5, 102, 400), (8, 106, 150), (5, 107, 1300)])
def test_random_ticks_through_the_c_abi(emulated_engine, oracle_lib, n_members, seed, groups):
    """(5, 107, 1300) carries >= 4096 messages per round: rgb_submit takes the class-dispatch kernel."""
    G.test_hip_equals_oracle_on_random_ticks(emulated_engine, oracle_lib, n_members, seed, groups)


def test_sub_ticks_ring_and_overflow(emulated_engine, oracle_lib):
    G.test_same_server_messages_are_serialised_in_submission_order(emulated_engine, oracle_lib)
    G.test_pipelined_ring_keeps_batches_in_order(emulated_engine, oracle_lib)
    G.test_collect_view_hands_out_the_slot_in_place(emulated_engine, oracle_lib)
    G.test_run_table_overflow_is_flagged(emulated_engine)
    G.test_leaderboard_snapshot(emulated_engine)
    for n_run0 in (3, 1, 2):
        G.test_write_below_first_index_keeps_the_range_start(emulated_engine, oracle_lib, n_run0)
    G.test_write_that_ends_below_a_sparse_range_leaves_no_range(emulated_engine, oracle_lib)


def test_big_batches_through_the_c_abi(emulated_engine, oracle_lib):
    """(the emulation has one stream: what this checks on the CPU is the enqueue logic and the ring with three big batches in flight)"""
    G.test_big_batches_pipelined_through_the_copy_stream(emulated_engine, oracle_lib, ticks=3)


def test_rounds_as_one_train_launch_through_the_c_abi(emulated_engine, oracle_lib):
    G.test_rounds_of_one_batch_run_as_one_train_launch(emulated_engine, oracle_lib, 6, G=1200, N=5, batches=2)
    G.test_rounds_of_one_batch_run_as_one_train_launch(emulated_engine, oracle_lib, 16, G=1200, N=5, batches=1)
    G.test_rounds_of_one_batch_run_as_one_train_launch(emulated_engine, oracle_lib, 6, G=1400, N=3, batches=1)


def test_concurrent_producers_and_consumers_through_the_c_abi(emulated_engine, oracle_lib):
    G.test_concurrent_producers_and_consumers_on_one_context(emulated_engine, oracle_lib, G=64, N=5, P=4, C_=2, per=5)


def test_wal_down_host_recipe_through_the_c_abi(emulated_engine, oracle_lib):
    G.test_wal_down_host_recipe_keeps_last_applied_and_the_log(emulated_engine, oracle_lib)


def test_wal_down_conditions_through_the_c_abi(emulated_engine, oracle_lib):
    for n, seed in ((3, 71), (5, 72), (7, 73), (1, 74)):
        G.test_wal_down_conditions_follower_and_leader_match_oracle(emulated_engine, oracle_lib, n, seed, groups=96, ticks=4)
    # >= 4096 messages per round: the class-dispatch kernel; then the rounds of one batch as ONE train launch
    G.test_wal_down_conditions_follower_and_leader_match_oracle(emulated_engine, oracle_lib, 5, 75, groups=1400, ticks=1)
    G.test_rounds_of_one_batch_run_as_one_train_launch(emulated_engine, oracle_lib, 6, G=1200, N=5, batches=1,
                                                       wal_down_share=0.25)
    G.test_leader_wal_down_host_recipe(emulated_engine, oracle_lib)


def test_sparse_pending_through_the_c_abi(emulated_engine, oracle_lib):
    for seed in range(8):
        G.test_sparse_pending_and_two_range_written_events(emulated_engine, oracle_lib, seed)
    G.test_written_event_for_a_live_index_below_the_snapshot(emulated_engine, oracle_lib)


def test_quorum_term_gate_on_any_run_table(emulated_engine, oracle_lib):
    G.test_quorum_term_gate_on_any_run_table(emulated_engine, oracle_lib, 4200, 231)
    G.test_quorum_term_gate_on_any_run_table(emulated_engine, oracle_lib, 64, 232)


def test_bounded_run_tables_and_repair_workload(emulated_engine, oracle_lib):
    G.test_bounded_run_table_matches_oracle(emulated_engine, oracle_lib, 5, 211, 300, 4)
    G.test_config5_log_matching_repair_matches_oracle(emulated_engine, oracle_lib)


def test_closed_loop_stream_through_the_c_abi(emulated_engine, oracle_lib):
    import test_cluster_safety as CS
    # the GPU test builds its own engine from ra_amd.engine, which is bound to the emulated library here
    CS.test_gpu_gives_identical_decisions_on_closed_loop_streams(oracle_lib, 5, 14, True)
    CS.test_gpu_gives_identical_decisions_on_closed_loop_streams(oracle_lib, 5, 15, "wal_down")   # WAL outages: both conditions


def test_leaderboard_allgather_host_gives_up_on_a_missing_rank(emulated_engine, monkeypatch):
    """rgb_leaderboard_allgather_host (what the NIF hands out as a binary) when a rank never arrives: the call returns
    RGB_E_COMM with the reason, the communicator is left aborted (the next call says so), and the caller's buffer is
    untouched -- the copy-outs land in pinned memory owned by the context and reach the caller only behind a successful
    wait (round-5 advisor finding).  On the CPU build the wait's time-out is played by RGB_EMU_LB_TIMEOUT; the real
    RCCL form needs two GPUs."""
    import ctypes as C
    engine = emulated_engine
    G, N = 24, 3
    eng = engine.RaGpuBatch(G, N, max_runs=8, ring_slots=1, ring_capacity=64)
    L = engine.lib()
    L.rgb_leaderboard_allgather_host.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    comm = engine.Comm(eng, engine.comm_unique_id(), 1, 0)
    rows = np.zeros(G, dtype=abi.LEADERBOARD_DTYPE)
    assert L.rgb_leaderboard_allgather_host(eng._h, comm.h, G, rows.ctypes.data) == 0          # the healthy path, one rank
    want = eng.snapshot()
    assert rows.tobytes() == want.tobytes()
    sentinel = np.full(G * abi.LEADERBOARD_DTYPE.itemsize, 0xA5, dtype=np.uint8)
    monkeypatch.setenv("RGB_EMU_LB_TIMEOUT", "1")
    rc = L.rgb_leaderboard_allgather_host(eng._h, comm.h, G, sentinel.ctypes.data)
    assert rc == abi.E_COMM, rc
    assert b"did not arrive" in L.rgb_comm_last_error() or b"timed out" in L.rgb_comm_last_error()
    assert (sentinel == 0xA5).all(), "a timed-out all-gather wrote into the caller's buffer"
    monkeypatch.delenv("RGB_EMU_LB_TIMEOUT")
    rc = L.rgb_leaderboard_allgather_host(eng._h, comm.h, G, sentinel.ctypes.data)
    assert rc == abi.E_COMM and b"aborted" in L.rgb_comm_last_error()
    assert (sentinel == 0xA5).all()
    comm.close()
    eng.close()
