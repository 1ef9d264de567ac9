#!/usr/bin/env python
"""fit_restarts on the C2 / C3 stand-ins: wall clock of n_init = 20 restarts for n_jobs in JOBS
(default 1,4,8); run under GPU_MAX_HW_QUEUES=... to change the number of hardware queues the streams
are mapped on.  (profiles/round3_restarts_threads.txt: more worker threads are not faster.)"""
import os
import sys
import time
import warnings

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd"))
sys.path.insert(0, _R)
import numpy as np  # noqa: E402
import convex_dim_red as cdr  # noqa: E402
from convex_dim_red import _backend  # noqa: E402
from oracle import aa_oracle as orc  # noqa: E402

warnings.simplefilter("ignore")
n_init = int(sys.argv[1]) if len(sys.argv) > 1 else 20
jobs_list = [int(j) for j in os.environ.get("JOBS", "1,4,8").split(",")]
graph = 0


def c2():
    n, p, k = 1610, 25000, 5
    rng = np.random.RandomState(0)
    B = rng.standard_normal((k, p))
    Zt = orc.right_stochastic_matrix((n, k), rng) ** 4
    Zt /= Zt.sum(axis=1, keepdims=True)
    return Zt.dot(B) + 0.05 * rng.standard_normal((n, p))


def c3():
    n, p, k = 22280, 167, 10
    rng = np.random.RandomState(0)
    W0 = rng.standard_normal((p, k))
    Zt = orc.right_stochastic_matrix((n, k), rng)
    return Zt.dot(W0.T) + 0.1 * rng.standard_normal((n, p))


for name, X, make in (
        ("C2", c2(), lambda rs: cdr.ArchetypalAnalysis(5, init="random", tolerance=1e-4, max_iterations=10000, random_state=rs,
                                                       dictionary_solver_kwargs=dict(max_iterations=1))),
        ("C3", c3(), lambda rs: cdr.GPNHConvexCoding(10, lambda_W=0, init="random", tolerance=1e-6, max_iterations=10000,
                                                     random_state=rs, stopping_criterion="rel_delta_f",
                                                     weights_solver_kwargs=dict(max_iterations=1)))):
    ref = None
    for jobs in jobs_list:
        for rep in range(2):
            shared = np.random.RandomState(0)
            t0 = time.perf_counter()
            models, best = cdr.fit_restarts(lambda: make(shared), X, n_init, n_jobs=jobs)
            t = time.perf_counter() - t0
        costs = [m.cost for m in models]
        iters = sum(m.n_iter + 1 for m in models)
        if ref is None:
            ref = costs
        print("%s use_graph=%d hw_queues=%s n_jobs=%d: %.3f s (%d outer iterations, %.0f it/s), costs identical to n_jobs=%d: %s, best %d"
              % (name, graph, os.environ.get("GPU_MAX_HW_QUEUES", "default"), jobs, t, iters, iters / t, jobs_list[0],
                 costs == ref, best), flush=True)
