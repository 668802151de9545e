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


# ---------------------------------------------------------------------------------------------------------------
# Sparse `pending`: after ra_log:write_sparse/3 + install_snapshot (src/ra_log.erl:601-635) the
# live indexes written below the snapshot are still unconfirmed, so `pending` is a ra_seq with gaps; written events
# carry sequences of one or two ranges.  The checker keeps the sequence as an explicit index list, the model as the
# reference's own high -> low list of indexes and ranges.

def _ranges(idxs):
    out = []
    for i in idxs:
        if out and out[-1][1] + 1 == i:
            out[-1][1] = i
        else:
            out.append([i, i])
    return out


def sparse_state(st, snap_idx, snap_term, live):
    """The host's re-upload after install_snapshot: empty range above the snapshot, the unconfirmed live indexes
    (at most two runs of them) as the old ranges of `pending`."""
    s = st.copy()
    s["snapshot_index"], s["snapshot_term"] = snap_idx, snap_term
    s["last_index"], s["last_term"] = snap_idx, snap_term
    s["first_index"] = snap_idx + 1
    s["last_written_index"], s["last_written_term"] = snap_idx, snap_term
    s["commit_index"], s["last_applied"] = snap_idx, snap_idx
    s["n_runs"] = 0
    s["run_start"] = 0
    s["run_term"] = 0
    s["current_term"] = snap_term
    s["pending_first"] = snap_idx + 1
    r = _ranges(live)
    assert len(r) <= 2
    s["n_pending_old"] = len(r)
    s["pending_old"] = 0
    for k, (a, b) in enumerate(r):
        s["pending_old"][0, k] = (a, b)
    return s


def pending_of(st):
    li = int(st["last_index"])
    out = []
    for k in range(int(st["n_pending_old"])):
        out += list(range(int(st["pending_old"][k][0]), int(st["pending_old"][k][1]) + 1))
    if int(st["first_index"]) <= li:
        out += list(range(int(st["pending_first"]), li + 1))
    return out


def sparse_history(step_fn, get_state, set_state, seed, steps=200, multi=False, stats=None):
    """Drives `step_fn(msg) -> decision` (the checker, or checker + engine in lock step) and the ra_seq model through
    one random history; yields nothing, asserts equality of last_written, pending and the resend / crash decisions."""
    from ra_log_model import LogModel, seq_expand, seq_from_list
    rng = np.random.default_rng(5000 + seed)
    snap_idx, term = int(rng.integers(20, 40)), int(rng.integers(1, 4))
    picks = sorted(set(int(x) for x in rng.integers(1, snap_idx, size=int(rng.integers(1, 6)))))
    live = []
    for r in _ranges(picks)[:2]:
        live += list(range(r[0], r[1] + 1))
    st = get_state()
    st[1] = sparse_state(st[1:2], snap_idx, term, live)[0]
    set_state(st)
    model = LogModel()
    model.range, model.terms, model.snap, model.lw = None, {}, (snap_idx, term), (snap_idx, term)
    model.last_term = term
    model.pending = seq_from_list(live)
    n_two, n_resend, n_clause2, n_multi = 0, 0, 0, 0
    for step in range(steps):
        st1 = get_state()[1]
        li, lt = int(st1["last_index"]), int(st1["last_term"])
        assert (li, lt) == model.last_index_term(), f"step {step}"
        assert pending_of(st1) == seq_expand(model.pending), f"step {step}: pending before"
        r = rng.random()
        if r < 0.30:                                        # append at the tail
            if rng.random() < 0.15:
                term += 1
            n = int(rng.integers(1, 4))
            d = step_fn(_msg(abi.MSG_AER, frm=0, term=term, a=li, b=lt, c=int(st1["commit_index"]), n_entries=n,
                             n_run0=n, run0_term=term))
            assert not (int(d["flags"][0]) & abi.F_INVARIANT), f"step {step}"
            model.write([(li + 1 + k, term) for k in range(n)])
        elif r < 0.38 and model.range and model.range[1] - max(model.range[0], int(st1["last_applied"])) >= 2:
            lo = max(model.range[0], int(st1["last_applied"])) + 1       # overwrite the tail
            fst = int(rng.integers(lo, model.range[1] + 1))
            prev_t = model.fetch_term(fst - 1)
            if prev_t is None:
                continue
            term += 1
            n = int(rng.integers(1, 3))
            d = step_fn(_msg(abi.MSG_AER, frm=0, term=term, a=fst - 1, b=prev_t, c=0, n_entries=n, n_run0=n,
                             run0_term=term))
            assert int(d["flags"][0]) & abi.F_WROTE
            model.write([(fst + k, term) for k in range(n)])
        elif r < 0.44 and model.range and model.range[1] - max(model.range[0], int(st1["last_applied"])) >= 1:
            lo = max(model.range[0], int(st1["last_applied"]))           # truncation: ra_log:set_last_index/2
            idx = int(rng.integers(lo, model.range[1]))
            t = model.fetch_term(idx)
            if t is None:
                continue
            term += 1
            d = step_fn(_msg(abi.MSG_AER, frm=0, term=term, a=idx, b=t, c=0, n_entries=0))
            assert int(d["flags"][0]) & abi.F_TRUNCATED
            assert model.set_last_index(idx)
        elif r < 0.92:                                      # a written event of one or two ranges
            pend = seq_expand(model.pending)
            runs = _ranges(pend)
            mode = rng.random()
            if len(runs) >= 2 and mode < 0.45:              # the two lowest pending runs (or prefixes of them)
                lo_r, hi_r = runs[0], runs[1]
                w_lo = list(range(lo_r[0], lo_r[1] + 1))
                w_hi = list(range(hi_r[0], int(rng.integers(hi_r[0], hi_r[1] + 1)) + 1))
            elif runs and mode < 0.75:                      # (a prefix of) the lowest run only
                w_lo, w_hi = [], list(range(runs[0][0], int(rng.integers(runs[0][0], runs[0][1] + 1)) + 1))
            elif len(runs) >= 2 and mode < 0.85:            # skips the lowest run: not a prefix
                w_lo, w_hi = [], list(range(runs[1][0], runs[1][1] + 1))
            else:                                           # anywhere
                b = max(1, li + int(rng.integers(-6, 2))); a = max(1, b - int(rng.integers(0, 4)))
                w_lo, w_hi = [], list(range(a, b + 1))
                if a > 3 and rng.random() < 0.5:
                    w_lo = list(range(max(1, a - 3 - int(rng.integers(0, 3))), a - 1))
            if w_lo and w_lo[-1] + 1 >= w_hi[0]:
                w_lo = []
            # (round 5) sequences of MORE than two ranges (RGB_MF_SEQX: the lower ones ride in the batch's range list):
            # every pending run (the top one maybe only a prefix), optionally one run split by a hole (not a prefix:
            # the reference re-sends), optionally one or two ranges of indexes that are not pending any more below them
            extra = []
            if multi and runs and rng.random() < 0.5:
                parts = [list(range(a, b + 1)) for a, b in runs]
                parts[-1] = parts[-1][:int(rng.integers(1, len(parts[-1]) + 1))]
                if rng.random() < 0.3:
                    k = int(rng.integers(0, len(parts)))
                    if len(parts[k]) >= 3:
                        h = int(rng.integers(1, len(parts[k]) - 1))
                        parts[k:k + 1] = [parts[k][:h], parts[k][h + 1:]]
                lowest = parts[0][0]
                junk = []
                for _ in range(int(rng.integers(0, 3))):
                    hi = lowest - 2 - int(rng.integers(0, 2))
                    lo = hi - int(rng.integers(0, 3))
                    if lo >= 1:
                        junk.insert(0, list(range(lo, hi + 1))); lowest = lo
                parts = junk + parts
                if len(parts) >= 3:
                    extra, w_lo, w_hi = parts[:-2], parts[-2], parts[-1]
            top = w_hi[-1]
            wt = model.fetch_term(min(top, li)) if model.range else None
            if wt is None or rng.random() < 0.15:
                wt = max(0, lt - int(rng.integers(0, 2)))
            kw = dict(term=wt, a=w_hi[0], b=w_hi[-1])
            if w_lo:
                kw.update(flags=abi.MF_SEQ2, run0_term=w_lo[0], run1_term=w_lo[-1])
                n_two += 1
            if extra:
                # the list carries an unrelated entry in front: the record's c is an offset into it
                lst = np.array([[1, 1]] + [[e[0], e[-1]] for e in extra], dtype=np.uint64)
                kw.update(flags=abi.MF_SEQ2 | abi.MF_SEQX, c=1, n_entries=len(extra))
                d = step_fn(_msg(abi.MSG_WRITTEN, **kw), seq_ranges=lst)
                n_multi += 1
            else:
                d = step_fn(_msg(abi.MSG_WRITTEN, **kw))
            seq = seq_from_list([i for e in extra for i in e] + w_lo + w_hi)
            if int(d["flags"][0]) & abi.F_INVARIANT:
                assert int(d["invariant"][0]) == abi.INV_WRITTEN_NOT_PREFIX
                with pytest.raises(AssertionError):
                    model.written(wt, seq)
                return n_two, n_resend, n_clause2, True     # the reference process would have crashed
            before = seq_expand(model.pending)
            lw_before = model.lw
            model.written(wt, seq)
            assert bool(int(d["flags"][0]) & abi.F_RESEND_PENDING) == model.resend, f"step {step}"
            n_resend += model.resend
            n_clause2 += (not model.resend and model.lw == lw_before and seq_expand(model.pending) != before)
        else:                                               # snapshot at last_applied
            la = int(st1["last_applied"])
            t = model.fetch_term(la)
            if t is None or la == 0:
                continue
            step_fn(_msg(abi.MSG_SNAPSHOT_WRITTEN, a=la, b=t))
            model.snapshot_written(la, t)
        st1 = get_state()[1]
        assert (int(st1["last_written_index"]), int(st1["last_written_term"])) == model.lw, f"step {step}"
        assert pending_of(st1) == seq_expand(model.pending), f"step {step}: pending {pending_of(st1)[:8]} vs {seq_expand(model.pending)[:8]}"
        if stats is not None:
            stats["multi"] = n_multi
    return n_two, n_resend, n_clause2, False


@pytest.mark.parametrize("seed", list(range(40)))
def test_oracle_sparse_pending_matches_ra_seq_model(oracle_lib, seed):
    cpu = oracle_lib.Oracle(1, 3)
    stats = sparse_history(lambda m: cpu.step(m)[0], cpu.get_state, lambda st: cpu.set_state(0, st), seed)
    cpu.close()
    assert stats is not None


@pytest.mark.parametrize("seed", list(range(40)))
def test_oracle_written_events_of_many_ranges_match_the_ra_seq_model(oracle_lib, seed):
    """(round 5) written events whose ra_seq has three to six ranges (RGB_MF_SEQX + the batch's range list) through the
    checker against the literal ra_seq model: same last_written, same pending, resend / crash where the model has them."""
    cpu = oracle_lib.Oracle(1, 3)
    stats = {}
    sparse_history(lambda m, seq_ranges=None: cpu.step(m, seq_ranges=seq_ranges)[0], cpu.get_state,
                   lambda st: cpu.set_state(0, st), 700 + seed, multi=True, stats=stats)
    cpu.close()


def test_many_range_histories_do_produce_many_range_events(oracle_lib):
    n = 0
    for seed in range(40):
        cpu = oracle_lib.Oracle(1, 3)
        stats = {}
        sparse_history(lambda m, seq_ranges=None: cpu.step(m, seq_ranges=seq_ranges)[0], cpu.get_state,
                       lambda st: cpu.set_state(0, st), 700 + seed, multi=True, stats=stats)
        cpu.close()
        n += stats.get("multi", 0)
    assert n > 60, n


def test_sparse_histories_cover_the_interesting_cases(oracle_lib):
    tot = np.zeros(4, dtype=np.int64)
    for seed in range(40):
        cpu = oracle_lib.Oracle(1, 3)
        tot += np.array(sparse_history(lambda m: cpu.step(m)[0], cpu.get_state, lambda st: cpu.set_state(0, st), seed),
                        dtype=np.int64)
        cpu.close()
    n_two, n_resend, n_clause2, crashed = tot
    assert n_two > 100 and n_resend > 20 and n_clause2 > 20, tot
