"""Random stochastic matrices (reference src/convex_dim_red/stochastic_matrices.py:15-39).

Stays host NumPy on purpose: the draws come from the caller's legacy ``RandomState``
and their order is part of the reproducibility contract."""
import numpy as np
from sklearn.utils import check_random_state


def _normalised_uniform(shape, random_state, axis):
    rng = check_random_state(random_state)
    draws = rng.uniform(size=shape)
    totals = draws.sum(axis=axis, keepdims=True)
    if axis not in (0, 1, -1, -2):
        raise ValueError("axis %d is out of bounds for array of dimension %d" % (axis, draws.ndim))
    return draws / totals


def left_stochastic_matrix(shape, random_state=None):
    """Random matrix with unit column sums."""
    return _normalised_uniform(shape, random_state, 0)


def right_stochastic_matrix(shape, random_state=None):
    """Random matrix with unit row sums."""
    return _normalised_uniform(shape, random_state, 1)
