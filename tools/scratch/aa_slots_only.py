#!/usr/bin/env python
"""Only the side-by-side fit of 12 AA restarts on the C2 stand-in (for profiling: tools/gpu_slots_prof.sh)."""
import os
import sys
import warnings

import numpy as np

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd"))
sys.path.insert(0, _R)
import convex_dim_red as cdr  # noqa: E402
from convex_dim_red import restarts  # noqa: E402
from oracle import aa_oracle as orc  # noqa: E402

warnings.simplefilter("ignore")
n, p, k = 1610, 25000, 5
rng = np.random.RandomState(0)
B = rng.standard_normal((k, p))
Zt = orc.right_stochastic_matrix((n, k), rng) ** 4
Zt /= Zt.sum(axis=1, keepdims=True)
X = Zt.dot(B) + 0.05 * rng.standard_normal((n, p))
shared = np.random.RandomState(0)
models, best = cdr.fit_restarts(lambda: cdr.ArchetypalAnalysis(k, init="random", tolerance=1e-4, max_iterations=10000,
                                                               random_state=shared,
                                                               dictionary_solver_kwargs=dict(max_iterations=1)), X, 12)
print(restarts.slots_profile, [m.n_iter for m in models])
