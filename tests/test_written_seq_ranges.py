"""Written events whose ra_seq has MORE than two ranges on the device (RGB_MF_SEQX + rgb_submit_seq; SURVEY 8(f) #2,
src/ra_log.erl:897-944, src/ra_seq.erl:17-66): the engine in lock step with the checker and with the literal ra_seq
model of tests/ra_log_model.py -- same decisions, same last_written, same pending set, RGB_F_RESEND_PENDING where the
reference re-sends, the remove_prefix badmatch as an invariant.  On the CPU block emulation and on the GPU (-m gpu).
Also: a record that names entries its list does not hold is refused (RGB_E_INVAL) and, on the device-resident path,
commits nothing (RGB_INV_WRITTEN_SEQ_LIST)."""
import numpy as np
import pytest

import test_pending_model as PM
from ra_amd import abi


def check(engine, oracle_lib, seed):
    cpu = oracle_lib.Oracle(1, 3)
    stats = {}
    with engine.RaGpuBatch(1, 3, ring_capacity=64, ring_slots=2, max_runs=16) as gpu:
        def step(m, seq_ranges=None):
            do, ro = cpu.step(m, seq_ranges=seq_ranges)
            dg, rg = gpu.step(m, seq_ranges=seq_ranges)
            assert dg.tobytes() == do.tobytes(), f"decision: engine {dg} checker {do} msg {m} list {seq_ranges}"
            assert gpu.get_state().tobytes() == cpu.get_state().tobytes(), "state"
            return do

        def set_state(st):
            cpu.set_state(0, st)
            gpu.set_state(0, st)
        PM.sparse_history(step, cpu.get_state, set_state, 700 + seed, steps=120, multi=True, stats=stats)
    cpu.close()
    return stats.get("multi", 0)


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_many_range_written_events_on_the_block_emulation(emulated_engine, oracle_lib, seed):
    check(emulated_engine, oracle_lib, seed)


def test_the_emulated_histories_contain_many_range_events(emulated_engine, oracle_lib):
    assert sum(check(emulated_engine, oracle_lib, s) for s in range(6, 12)) > 8


def test_a_record_outside_its_list_is_refused(emulated_engine):
    with emulated_engine.RaGpuBatch(1, 3, ring_capacity=64, ring_slots=2) as eng:
        m = PM._msg(abi.MSG_WRITTEN, term=1, a=20, b=22, flags=abi.MF_SEQ2 | abi.MF_SEQX, run0_term=10, run1_term=12, c=1, n_entries=2)
        lst = np.array([[1, 2], [4, 5]], dtype=np.uint64)
        with pytest.raises(emulated_engine.RgbError):
            eng.submit(m, seq_ranges=lst)                   # entries 1..2 of a two-entry list
        with pytest.raises(emulated_engine.RgbError):
            eng.submit(m)                                   # no list at all
        m["c"] = 0
        eng.submit(m, seq_ranges=lst)                       # fine
        eng.collect(cap=1)
        with pytest.raises(emulated_engine.RgbError):
            eng.submit(m, seq_ranges=np.array([[1, 2], [3, 5]], dtype=np.uint64))     # adjacent ranges are one range


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5, 6, 7])
def test_many_range_written_events_on_the_gpu(oracle_lib, seed):
    import os
    from ra_amd import engine
    if not os.path.exists(engine.LIB_PATH):
        engine.build()
    engine.lib()
    check(engine, oracle_lib, seed)
