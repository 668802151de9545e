"""A third, independent restatement of the ra_log cursor rules the hot path relies on -- plain
Python, literal about the reference's data shapes: the log is a dict index -> term, `pending` is a
real ra_seq (a high -> low list of indexes and {low, high} ranges, src/ra_seq.erl:8-12) handled by
restatements of ra_seq:append/2, limit/2, floor/2, remove_prefix/2 (src/ra_seq.erl:44-110,
144-160, 278-311), the range by ra_range:limit/2 and truncate/2 (src/ra_range.erl:80-106).
TEST INFRASTRUCTURE: used only by tests/test_pending_model.py to cross-check oracle/ra_oracle.c,
which keeps `pending` as one integer."""


# ---- ra_seq (src/ra_seq.erl) -------------------------------------------------------------
def seq_append(idx, seq):                       # append/2 :44-66
    if len(seq) >= 2 and isinstance(seq[0], int) and isinstance(seq[1], int) and \
            idx == seq[0] + 1 and idx == seq[1] + 2:
        return [(seq[1], idx)] + seq[2:]
    if seq and isinstance(seq[0], tuple) and idx == seq[0][1] + 1:
        return [(seq[0][0], idx)] + seq[1:]
    if not seq:
        return [idx]
    top = seq[0][1] if isinstance(seq[0], tuple) else seq[0]
    assert idx > top, "ra_seq:append/2 function_clause"
    return [idx] + seq


def seq_expand(seq):                            # ascending list of indexes
    out = []
    for e in reversed(seq):
        out.extend(range(e[0], e[1] + 1) if isinstance(e, tuple) else [e])
    return out


def seq_from_list(idxs):
    s = []
    for i in sorted(set(idxs)):
        s = seq_append(i, s)
    return s


def seq_limit(ceil_incl, seq):                  # limit/2 :81-98
    return seq_from_list([i for i in seq_expand(seq) if i <= ceil_incl])


def seq_floor(floor_incl, seq):                 # floor/2 :74-79, 293-311
    return seq_from_list([i for i in seq_expand(seq) if i >= floor_incl])


def seq_remove_prefix(prefix, seq):             # remove_prefix/2 :144-147, drop_prefix/2 :278-291
    p, s = seq_expand(prefix), seq_expand(seq)
    pi = si = 0
    while True:
        if si >= len(s):
            return True, []                     # drop_prefix(_, end_of_seq) -> {ok, []}
        if pi >= len(p):
            return True, seq_from_list(s[si:])  # prefix exhausted: the rest stays
        if p[pi] == s[si]:
            pi += 1; si += 1
        elif p[pi] < s[si]:
            pi += 1                             # prefix index below the sequence: skipped
        else:
            return False, None                  # {error, not_prefix}


# ---- ra_log cursors (src/ra_log.erl) -----------------------------------------------------------
class LogModel:
    """range = (first, last) or None; terms: index -> term inside the range."""

    def __init__(self):
        self.range = (0, 0)
        self.terms = {0: 0}
        self.last_term = 0
        self.lw = (0, 0)
        self.snap = None
        self.pending = []
        self.resend = False

    def fetch_term(self, idx):                  # :1186-1200
        if self.range and self.range[0] <= idx <= self.range[1]:
            return self.terms[idx]
        return None

    def last_index_term(self):                  # :830-835
        if self.range:
            return self.range[1], self.last_term
        return self.snap if self.snap else (None, None)

    def write(self, entries):                   # write/2 :547-599 + wal_write_batch :1596-1629
        fst = entries[0][0]
        assert self.range is None or (0 <= fst <= self.range[1] + 1)
        lwi = min(fst - 1, self.lw[0])
        if lwi == self.lw[0]:
            lwt = self.lw[1]
        elif self.snap and self.snap[0] == lwi:
            lwt = self.snap[1]
        elif lwi <= 0:
            lwt = 0
        else:
            lwt = self.fetch_term(lwi)
            assert lwt is not None
        pend = seq_limit(fst - 1, self.pending)
        for idx, term in entries:
            self.terms[idx] = term
            pend = seq_append(idx, pend)
        last = entries[-1][0]
        for k in [k for k in self.terms if k > last]:
            del self.terms[k]
        self.range = (self.range[0] if self.range else fst, last)
        self.last_term = entries[-1][1]
        self.pending = pend
        self.lw = (lwi, lwt)

    def set_last_index(self, idx):              # :842-893
        t = self.fetch_term(idx)
        cur = self.snap
        if t is None and not (cur and cur[0] == idx):
            return False
        if cur and cur[0] == idx:
            self.range = self._limit(idx + 1)
            self.last_term = cur[1]
            self.pending = seq_limit(idx, self.pending)
            self.lw = cur
            self._prune()
            return True
        lwi = min(idx, self.lw[0])
        lwt = cur[1] if (cur and cur[0] == lwi) else self.fetch_term(lwi)
        assert lwt is not None
        self.range = self._limit(idx + 1)
        self.last_term = t
        self.pending = seq_limit(idx, self.pending)
        self.lw = (lwi, lwt)
        self._prune()
        return True

    def _limit(self, ceil_excl):                # ra_range:limit/2
        if self.range is None:
            return None
        s, e = self.range
        if ceil_excl <= s:
            return None
        return (s, ceil_excl - 1) if ceil_excl <= e else (s, e)

    def _prune(self):
        if self.range is None:
            self.terms = {}
        else:
            self.terms = {k: v for k, v in self.terms.items() if self.range[0] <= k <= self.range[1]}

    def written(self, term, seq):               # handle_event({written, ..}) :897-944
        self.resend = False
        while True:
            last = seq[0][1] if isinstance(seq[0], tuple) else seq[0]
            t = self.fetch_term(last)
            if t is not None and t == term:
                ok, pend = seq_remove_prefix(seq, self.pending)
                if ok:
                    self.lw = (last, term)
                    self.pending = pend
                else:
                    self.resend = True          # resend_pending/2: host I/O, cursors unchanged
                return
            if t is None and self.snap and last <= self.snap[0]:
                ok, pend = seq_remove_prefix(seq, self.pending)
                assert ok, "{ok, Pend} = ra_seq:remove_prefix(..) badmatch"
                self.pending = pend
                return
            seq = seq_limit(last - 1, seq)
            if not seq:
                return

    def snapshot_written(self, idx, term):      # :1054-1150, no live indexes
        if not (self.range and idx >= self.range[0]):
            return
        if not (self.lw[0] > idx):
            self.lw = (idx, term)
        self.pending = seq_floor(idx + 1, self.pending)
        s, e = self.range                       # ra_range:truncate/2
        self.range = None if idx >= e else (max(s, idx + 1), e)
        self.snap = (idx, term)
        self._prune()
