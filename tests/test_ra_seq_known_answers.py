"""The ra_seq restatement of tests/ra_log_model.py (which cross-checks the checker's one-integer
`pending`) pinned to the known answers of the reference's own test/ra_seq_SUITE.erl: append (:51-59),
floor (:61-74), limit (:76-90), remove_prefix (:160-174), remove_prefix_2 (:176-180),
from_list_with_duplicates (:182-199).  Erlang's {Low, High} ranges are (low, high) tuples here."""
import pytest

from ra_log_model import seq_append, seq_expand, seq_from_list, seq_limit, seq_floor, seq_remove_prefix


def test_append():
    s1 = seq_append(1, []); assert s1 == [1]
    s2 = seq_append(2, s1); assert s2 == [2, 1]
    s3 = seq_append(3, s2); assert s3 == [(1, 3)]
    s4 = seq_append(4, s3); assert s4 == [(1, 4)]
    s5 = seq_append(6, s4); assert s5 == [6, (1, 4)]
    with pytest.raises(AssertionError):                       # ?assertError(function_clause, ...)
        seq_append(2, s4)
    with pytest.raises(AssertionError):
        seq_append(6, s5)


def test_floor():
    s = seq_from_list([1, 2, 3, 5, 6, 7, 8, 9, 11])
    assert s == [11, (5, 9), (1, 3)]
    want = {11: [11], 9: [11, 9], 8: [11, 9, 8], 7: [11, (7, 9)], 6: [11, (6, 9)], 5: [11, (5, 9)], 4: [11, (5, 9)],
            3: [11, (5, 9), 3], 2: [11, (5, 9), 3, 2], 1: [11, (5, 9), (1, 3)], 0: [11, (5, 9), (1, 3)]}
    for k, v in want.items():
        assert seq_expand(seq_floor(k, s)) == seq_expand(v), k
    # the exact shapes too, except where the suite's answers keep a two-element run as two integers
    # ([11, 9, 8], [.., 3, 2]): append/2 only folds a run into a range from its third element on
    assert seq_floor(7, s) == [11, (7, 9)] and seq_floor(9, s) == [11, 9] and seq_floor(8, s) == [11, 9, 8]
    assert seq_floor(2, s) == [11, (5, 9), 3, 2]


def test_limit():
    s = seq_from_list([1, 2, 3, 5, 6, 7, 8, 9, 11])
    want = {11: [11, (5, 9), (1, 3)], 10: [(5, 9), (1, 3)], 9: [(5, 9), (1, 3)], 8: [(5, 8), (1, 3)],
            7: [(5, 7), (1, 3)], 6: [6, 5, (1, 3)], 5: [5, (1, 3)], 4: [(1, 3)], 3: [(1, 3)], 2: [2, 1], 1: [1], 0: []}
    for k, v in want.items():
        assert seq_limit(k, s) == v, k


def test_remove_prefix():
    s0 = seq_from_list([2, 3, 5, 6, 8, 9, 10, 12])
    ok, s1 = seq_remove_prefix(seq_from_list([2, 3, 5]), s0)
    assert ok and seq_expand(s1)[::-1] == [12, 10, 9, 8, 6]
    ok, s2 = seq_remove_prefix(seq_from_list([1, 2, 3, 5]), s0)       # prefix includes already removed items
    assert ok and seq_expand(s2)[::-1] == [12, 10, 9, 8, 6]
    ok, _ = seq_remove_prefix(seq_from_list([5, 6, 8]), s0)           # {error, not_prefix}
    assert not ok
    assert seq_remove_prefix(s0, s0) == (True, [])


def test_remove_prefix_2():
    assert seq_remove_prefix(seq_from_list([1, 2, 3]), seq_from_list([2, 3, 4, 5])) == (True, [5, 4])


def test_from_list_with_duplicates():
    assert seq_from_list([1, 2, 2, 3]) == [(1, 3)]
    assert seq_from_list([5, 5, 5, 5]) == [5]
    assert seq_from_list([3, 1, 2, 1, 3, 2]) == [(1, 3)]
    s4 = seq_from_list([1, 2, 3, 3, 5, 6, 7, 7, 10, 11, 11])
    assert s4 == [11, 10, (5, 7), (1, 3)]
    assert seq_expand(s4)[::-1] == [11, 10, 7, 6, 5, 3, 2, 1]
