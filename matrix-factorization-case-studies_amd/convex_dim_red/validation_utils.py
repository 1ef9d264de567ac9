"""Input validation helpers (reference src/convex_dim_red/validation_utils.py:11-35);
host-side, same messages and ValueError behaviour."""
import numpy as np


def check_unit_axis_sums(a, whom, axis=0):
    if not np.isclose(np.sum(a, axis=axis), 1).all():
        raise ValueError("Array with incorrect axis sums passed to %s. "
                         "Expected sums along axis %d to be 1." % (whom, axis))


def check_array_shape(a, shape, whom):
    if tuple(a.shape) != tuple(shape):
        raise ValueError("Array with wrong shape passed to %s. Expected %s, but got %s"
                         % (whom, shape, a.shape))


def check_stochastic_matrix(a, shape, whom, axis=0):
    check_array_shape(a, shape, whom)
    check_unit_axis_sums(a, whom, axis=axis)
