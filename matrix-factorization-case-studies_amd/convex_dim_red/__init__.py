"""convex_dim_red -- MI355X (gfx950) implementation of the archetypal-analysis /
GPNH-convex-coding solvers, keeping the reference package's module surface
(reference src/convex_dim_red/__init__.py:5-11) so driver scripts and tests switch
over by putting this directory on ``PYTHONPATH`` instead of the reference's ``src``.

All numerics run in hand-written HIP kernels behind the C ABI of ``libaa_hip.so``
(``include/aa_hip.h``); there is no CPU fallback.
"""
from .archetypal_analysis import ArchetypalAnalysis, KernelAA
from .furthest_sum import furthest_sum
from .gpnh_convex_coding import GPNHConvexCoding
from .kmeans import gap_statistic
from .simplex_projection import simplex_project_rows, simplex_project_columns
from .spg import spg
from .stochastic_matrices import left_stochastic_matrix, right_stochastic_matrix
from ._backend import release_device_cache
from .preprocessing import DeviceData, weight_and_flatten_on_device
from .restarts import fit_restarts

__all__ = ["ArchetypalAnalysis", "KernelAA", "GPNHConvexCoding", "furthest_sum",
           "gap_statistic", "simplex_project_rows", "simplex_project_columns", "spg",
           "left_stochastic_matrix", "right_stochastic_matrix", "release_device_cache",
           "DeviceData", "weight_and_flatten_on_device", "fit_restarts"]
