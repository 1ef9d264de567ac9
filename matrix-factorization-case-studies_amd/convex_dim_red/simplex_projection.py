"""Euclidean projection onto the unit simplex, on the GPU.

Mirrors reference src/convex_dim_red/simplex_projection.py: ``simplex_project_vector``
(:13-27), ``simplex_project_columns`` (:30-37), ``simplex_project_rows`` (:40-47).
The device kernels are sort-free (kernels_qp.hip: k_simplex_rows) but end at the same
support and the same closed-form threshold as the reference's sorted scan."""
import numpy as np

from . import _backend


def simplex_project_rows(A):
    """Project every row of a 2-D array; returns a new array."""
    A = np.asarray(A, dtype=np.float64)
    if A.ndim != 2:
        raise ValueError("simplex_project_rows expects a 2-D array")
    return _backend.simplex_project_rows(A)


def simplex_project_columns(A):
    """Project every column of a 2-D array; returns a new array."""
    A = np.asarray(A, dtype=np.float64)
    if A.ndim != 2:
        raise ValueError("simplex_project_columns expects a 2-D array")
    return np.ascontiguousarray(_backend.simplex_project_rows(np.ascontiguousarray(A.T)).T)


def simplex_project_vector(x):
    """Project a 1-D array."""
    x = np.asarray(x, dtype=np.float64)
    return _backend.simplex_project_rows(x.reshape(1, -1))[0]
