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
    """Candidate pool as two aligned arrays in the reference's list order."""

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
