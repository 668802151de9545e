"""Cross-check of the checker's one-integer `pending` (and of its last_written rules) against
tests/ra_log_model.py, which keeps `pending` as a real ra_seq and follows src/ra_log.erl and
src/ra_seq.erl literally.  Random follower histories: appends, overwrites, truncations, written
events (in order, late, overlapping, with gaps, with stale terms) and snapshots."""
import numpy as np
import pytest

from ra_amd import abi
from ra_log_model import LogModel, seq_expand


def _msg(kind, **kw):
    m = np.zeros(1, dtype=abi.MSG_DTYPE)
    m["server"] = 1
    m["kind"] = kind
    m["from"] = kw.pop("frm", abi.NONE)
    for k, v in kw.items():
        m[k] = v
    return m


@pytest.mark.parametrize("seed", list(range(24)))
def test_oracle_pending_matches_ra_seq_model(oracle_lib, seed):
    rng = np.random.default_rng(1000 + seed)
    cpu = oracle_lib.Oracle(1, 3)
    model = LogModel()
    term = 1
    for step in range(300):
        st = cpu.get_state()[1]
        li, lt = int(st["last_index"]), int(st["last_term"])
        assert (li, lt) == model.last_index_term(), f"step {step}"
        r = rng.random()
        if r < 0.35:                                        # append at the tail (sometimes a new term)
            if rng.random() < 0.15:
                term += 1
            n = int(rng.integers(1, 5))
            ents = [(li + 1 + k, term) for k in range(n)]
            commit = int(rng.integers(0, li + 1)) if rng.random() < 0.5 else 0
            d, _ = cpu.step(_msg(abi.MSG_AER, frm=0, term=term, a=li, b=lt, c=commit, n_entries=n, n_run0=n,
                                 run0_term=term))
            assert not (int(d["flags"][0]) & abi.F_INVARIANT)
            model.write(ents)
        elif r < 0.45 and model.range and model.range[1] - max(model.range[0], int(st["last_applied"])) >= 2:
            # overwrite the tail from a random index above last_applied with a new term
            lo = max(model.range[0], int(st["last_applied"])) + 1
            fst = int(rng.integers(lo, model.range[1] + 1))
            term += 1
            n = int(rng.integers(1, 4))
            prev_t = model.fetch_term(fst - 1)
            if prev_t is None:
                continue
            d, _ = cpu.step(_msg(abi.MSG_AER, frm=0, term=term, a=fst - 1, b=prev_t, c=0, n_entries=n, n_run0=n,
                                 run0_term=term))
            assert not (int(d["flags"][0]) & abi.F_INVARIANT)
            assert int(d["flags"][0]) & abi.F_WROTE
            model.write([(fst + k, term) for k in range(n)])
        elif r < 0.52 and model.range and model.range[1] - max(model.range[0], int(st["last_applied"])) >= 1:
            # a new leader's empty append_entries_rpc truncates the tail: ra_log:set_last_index/2
            lo = max(model.range[0], int(st["last_applied"]))
            idx = int(rng.integers(lo, model.range[1]))
            t = model.fetch_term(idx)
            if t is None:
                continue
            term += 1
            d, _ = cpu.step(_msg(abi.MSG_AER, frm=0, term=term, a=idx, b=t, c=0, n_entries=0))
            assert int(d["flags"][0]) & abi.F_TRUNCATED
            assert model.set_last_index(idx)
        elif r < 0.90:                                      # a written event
            pend = seq_expand(model.pending)
            mode = rng.random()
            if pend and mode < 0.55:                        # in order: a prefix of pending
                a, b = pend[0], pend[int(rng.integers(0, len(pend)))]
            elif pend and mode < 0.70:                      # a gap: starts above the first pending index
                a = pend[0] + int(rng.integers(1, 3)); b = a + int(rng.integers(0, 3))
            else:                                           # anywhere around the log
                b = max(0, li + int(rng.integers(-4, 2))); a = max(0, b - int(rng.integers(0, 5)))
            wt = model.fetch_term(min(b, li))
            if wt is None or rng.random() < 0.2:
                wt = max(0, lt - int(rng.integers(0, 2)))
            d, _ = cpu.step(_msg(abi.MSG_WRITTEN, term=wt, a=a, b=b))
            if int(d["flags"][0]) & abi.F_INVARIANT:
                assert int(d["invariant"][0]) == abi.INV_WRITTEN_NOT_PREFIX
                with pytest.raises(AssertionError):
                    model.written(wt, [(a, b)] if b > a else [a])
                break                                       # the reference process would have crashed
            model.written(wt, [(a, b)] if b > a else [a])
            assert bool(int(d["flags"][0]) & abi.F_RESEND_PENDING) == model.resend, f"step {step}"
        else:                                               # snapshot at last_applied / a bit beyond
            la = int(st["last_applied"])
            idx = la if rng.random() < 0.7 else min(li, la + int(rng.integers(0, 3)))
            t = model.fetch_term(idx)
            if t is None or idx == 0:
                continue
            cpu.step(_msg(abi.MSG_SNAPSHOT_WRITTEN, a=idx, b=t))
            model.snapshot_written(idx, t)
        st = cpu.get_state()[1]
        assert (int(st["last_written_index"]), int(st["last_written_term"])) == model.lw, f"step {step}"
        pend = seq_expand(model.pending)
        li = int(st["last_index"])
        got = list(range(int(st["pending_first"]), li + 1)) if int(st["first_index"]) <= li else []
        assert got == pend, f"step {step}: pending {got[:6]}.. vs ra_seq {pend[:6]}.."
    cpu.close()
