#!/usr/bin/env python
"""Row-local pass at the headline size: time and accuracy with the fp32 accumulation chain cut every
32 columns and summed in float64 (row_local_acc64 = 1) against the plain fp32 chain (0)."""
import os
import sys
import time

import numpy as np

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd"))
sys.path.insert(0, _R)
import bench  # noqa: E402
from convex_dim_red import _backend  # noqa: E402

n, p, k = int(os.environ.get("N", bench.N_SAMPLES)), bench.N_FEATURES, bench.N_COMPONENTS
X = bench.synthetic_rows(0, n)
C0, Z0 = bench.start_factors(n, k)
rows = np.arange(0, n, max(1, n // 2000))              # accuracy on a sample of the rows
B = C0.dot(X.astype(np.float64)) if n <= 30000 else X[rows[:k]].astype(np.float64)
want = X[rows].astype(np.float64).dot(B.T)
scale = np.abs(X[rows]).astype(np.float64).dot(np.abs(B).T)
with _backend.Context(dtype="float32") as ctx:
    ctx.set_data(X)
    for a64, waves in ((0, 0), (1, 0), (1, 12), (1, 11), (1, 10), (0, 12), (0, 0), (1, 0)):
        _backend.set_option("row_local_acc64", a64)
        _backend.set_option("row_local_waves", waves)
        got = ctx.pass_row_local(B)[rows]
        err = np.abs(got - want) / scale
        ctx.set_state(C0, Z0, np.ones(k))
        ctx.prepare()
        ctx.time_kernel(1, 5)
        ms = ctx.time_kernel(1, 40)
        print("acc64 %d waves %2d: %.4f ms  %.2f TB/s | err / sum|x||b|: rms %.2e max %.2e" %
              (a64, waves, ms, n * p * 4 / ms / 1e9, np.sqrt((err ** 2).mean()), err.max()), flush=True)
    _backend.set_option("row_local_waves", 0)
    for a64 in (0, 1, 0, 1):
        _backend.set_option("row_local_acc64", a64)
        ctx.set_state(C0, Z0, np.ones(k))
        ctx.prepare()
        ctx.outer_iterations(5, dict(max_iterations=1), {})
        t0 = time.perf_counter()
        ctx.outer_iterations(20, dict(max_iterations=1), {})
        dt = time.perf_counter() - t0
        print("acc64 %d: %.3f ms per outer iteration (iterations 5..25)" % (a64, 1e3 * dt / 20), flush=True)
_backend.set_option("row_local_acc64", 1)
