"""FurthestSum initialisation (Morup & Hansen 2012).

``furthest_sum`` mirrors reference src/convex_dim_red/furthest_sum.py:130-170 for a
caller-supplied dissimilarity matrix.  ``furthest_sum_from_columns`` is the same
selection rule driven by a callback that returns ONE column of the dissimilarity matrix
at a time -- on the hot path that callback is ``Context.distance_column`` (a GEMV
against the resident data matrix, csrc/kernels_tall.hip: k_distance_data), so the
n x n matrices the reference builds (archetypal_analysis.py:95-100) never exist.

Selection rule kept exactly (reference :17-20): the candidate pool is stably sorted by
running distance sum and its LAST element is taken, so among equal sums the candidate
positioned last in the pool wins, and the pool stays sorted for the next pick.
"""
import numpy as np


def _validate(n_samples, n_components, start_index, exclude):
    if start_index >= n_samples:
        raise ValueError("Start index %r is out of bounds (n_samples = %d)"
                         % (start_index, n_samples))
    for index in exclude:
        if index == start_index:
            raise ValueError("Start index %r is excluded" % start_index)
    n_excluded = len(exclude)
    if n_excluded < n_samples and n_components > n_samples - n_excluded:
        raise ValueError(
            "Too few point available to select requested number of components "
            "(n_components=%d, n_samples=%d, n_excluded=%d)"
            % (n_components, n_samples, n_excluded))


class _Pool(object):
    """The reference's candidate list without its sorts.  The reference stably sorts the list by running
    sum before every pick and pops the last element, so (a) the pick is the candidate with the
    largest sum and (b) among equal sums the one positioned last wins, where "position" is the order
    the PREVIOUS stable sort left -- i.e. the order by the sums at the previous pick, ties by the pick
    before, ..., finally by the initial order; a candidate pushed back in between sits at the end of
    the list (it compares as +inf at the sort before its push).  So the pick is a maximum search
    (O(n)) and the history of sum vectors is consulted only to break a tie -- instead of a stable
    sort of all candidates per pick (1.5 ms at 22 280 candidates, 19 picks per initialisation).
    ``_SortedPool`` below is the literal form; tests/test_cpu_host.py checks the two against each
    other on tie-heavy inputs."""

    def __init__(self, members, sums):
        members = np.asarray(members, dtype=np.int64)
        self.n = int(members.max()) + 1 if members.size else 0
        self.alive = np.zeros(self.n, dtype=bool)
        self.alive[members] = True
        self.sums = np.full(self.n, -np.inf)
        self.sums[members] = np.asarray(sums, dtype=np.float64)
        self.initial = np.full(self.n, -1, dtype=np.int64)       # position in the initial list
        self.initial[members] = np.arange(members.size)
        self.history = []                                         # sum vectors at the earlier sorts, latest last

    def _grow(self, n):
        if n <= self.n:
            return
        pad = n - self.n
        self.alive = np.concatenate((self.alive, np.zeros(pad, dtype=bool)))
        self.sums = np.concatenate((self.sums, np.full(pad, -np.inf)))
        self.initial = np.concatenate((self.initial, np.full(pad, -1, dtype=np.int64)))
        self.history = [np.concatenate((h, np.full(pad, -np.inf))) for h in self.history]
        self.n = n

    def take_furthest(self):
        cur = np.where(self.alive, self.sums, -np.inf)
        tied = np.flatnonzero(cur == cur.max())
        depth = len(self.history)
        while tied.size > 1 and depth > 0:                        # order the previous sorts left
            depth -= 1
            h = self.history[depth][tied]
            tied = tied[h == h.max()]
        if tied.size > 1:                                         # never told apart: the initial order
            tied = tied[[int(np.argmax(self.initial[tied]))]]
        chosen = int(tied[0])
        self.history.append(cur)                                  # the sort that has just "happened"
        self.alive[chosen] = False
        return chosen

    def shift(self, column, sign):
        column = np.asarray(column)
        self._grow(column.shape[0])
        self.sums[self.alive] += sign * column[:self.n][self.alive]

    def push(self, member, value):
        member = int(member)
        self._grow(member + 1)
        self.alive[member] = True
        self.sums[member] = value
        # appended at the end of the list: after every candidate in the order the latest sort left
        if self.history:
            self.history[-1] = self.history[-1].copy()
            self.history[-1][member] = np.inf
        else:
            self.initial[member] = self.initial.max() + 1


class _SortedPool(object):
    """Candidate pool as two aligned arrays in the reference's list order (the literal form:
    furthest_sum.py:17-20 of the reference)."""

    def __init__(self, members, sums):
        self.members = np.asarray(members, dtype=np.int64)
        self.sums = np.asarray(sums, dtype=np.float64).copy()

    def take_furthest(self):
        order = np.argsort(self.sums, kind="stable")
        self.members = self.members[order]
        self.sums = self.sums[order]
        chosen = int(self.members[-1])
        self.members = self.members[:-1]
        self.sums = self.sums[:-1]
        return chosen

    def shift(self, column, sign):
        self.sums += sign * column[self.members]

    def push(self, member, value):
        self.members = np.append(self.members, member)
        self.sums = np.append(self.sums, value)


def furthest_sum_from_columns(column_of, row_entry, n_samples, n_components, start_index,
                              exclude=None, extra_steps=1):
    """``column_of(j)`` returns d[:, j] as seen from the candidates (the reference reads
    ``D[new, i]`` when adding and ``D[i, old]`` when removing -- the two callbacks let
    an asymmetric matrix be honoured); ``row_entry(i, j)`` returns D[i, j]."""
    if n_components == 0:
        return []
    exclude = [] if exclude is None else list(exclude)
    _validate(n_samples, n_components, start_index, exclude)

    selected = np.full((n_components,), start_index)
    blocked = np.zeros(n_samples, dtype=bool)
    blocked[np.asarray(exclude, dtype=np.int64)] = True
    blocked[start_index] = True
    members = np.flatnonzero(~blocked)
    pool = _Pool(members, column_of(start_index, "into")[members])

    for slot in range(1, n_components):
        selected[slot] = pool.take_furthest()
        pool.shift(column_of(selected[slot], "from"), +1.0)

    for step in range(max(extra_steps, 0)):
        slot = step % n_components
        leaving = selected[slot]
        pool.shift(column_of(leaving, "into"), -1.0)
        back = 0
        for other in selected:
            if other != leaving:
                back += row_entry(leaving, other)
        pool.push(leaving, back)
        selected[slot] = pool.take_furthest()
        pool.shift(column_of(selected[slot], "from"), +1.0)
    return selected


def furthest_sum(dissimilarity_matrix, n_components, start_index,
                 exclude=None, extra_steps=1):
    """Select ``n_components`` far-apart samples from a dissimilarity matrix."""
    D = np.asarray(dissimilarity_matrix)
    if D.ndim != 2 or D.shape[0] != D.shape[1]:
        raise ValueError("Dissimilarity matrix must be square, but got shape %r"
                         % list(D.shape))

    def column_of(j, sense):
        # "into": D[i, j] (distance of candidate i to j); "from": D[j, i]
        return D[:, j] if sense == "into" else D[j, :]

    return furthest_sum_from_columns(column_of, lambda i, j: D[i, j], D.shape[0],
                                     n_components, start_index, exclude=exclude,
                                     extra_steps=extra_steps)
