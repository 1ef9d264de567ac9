#!/usr/bin/env python
"""Row-local pass at the headline size: the LDS-DMA kernel (row_local_variant 9, float64 sums of
32-column pieces) against the register-staged wave-streaming kernel (8): time and accuracy."""
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
rows = np.arange(0, n, max(1, n // 3000))
rng = np.random.RandomState(0)
B = X[rng.choice(n, k, replace=False)].astype(np.float64) * 0.7 + 0.3 * rng.standard_normal((k, p))
want = X[rows].astype(np.float64).dot(B.T)
scale = np.abs(X[rows]).astype(np.float64).dot(np.abs(B).T)
with _backend.Context(dtype="float32") as ctx:
    ctx.set_data(X)
    for variant, waves, ring, nt in ((8, 0, 0, 0), (9, 0, 0, 0), (9, 0, 8, 0), (9, 0, 9, 0), (9, 0, 10, 0), (9, 0, 0, 1), (9, 14, 0, 0), (9, 16, 0, 0), (8, 0, 0, 0), (9, 0, 0, 0)):
        _backend.set_option("row_local_variant", variant)
        _backend.set_option("row_local_waves", waves)
        _backend.set_option("row_local_ring", ring)
        _backend.set_option("row_local_nt", nt)
        full = ctx.pass_row_local(B)
        got = full[rows]
        err = np.abs(got - want) / scale
        ctx.set_state(C0, Z0, np.ones(k))
        ctx.prepare()
        ctx.time_kernel(1, 5)
        ms = ctx.time_kernel(1, 40)
        print("variant %d waves %2d ring %2d nt %d: %.4f ms  %.2f TB/s | err / sum|x||b|: rms %.2e max %.2e  (last rows ok: %s)" %
              (variant, waves, ring, nt, ms, n * p * 4 / ms / 1e9, np.sqrt((err ** 2).mean()), err.max(),
               bool(np.isfinite(full[-40:]).all())), flush=True)
    _backend.set_option("row_local_waves", 0)
    _backend.set_option("row_local_ring", 8)
    _backend.set_option("row_local_nt", 0)
    for variant in (8, 9, 8, 9):
        _backend.set_option("row_local_variant", variant)
        ctx.set_state(C0, Z0, np.ones(k))
        ctx.prepare()
        ctx.outer_iterations(5, dict(max_iterations=1), {})
        t0 = time.perf_counter()
        c = ctx.outer_iterations(20, dict(max_iterations=1), {})
        dt = time.perf_counter() - t0
        print("variant %d: %.3f ms per outer iteration (iterations 5..25), cost %.9f" % (variant, 1e3 * dt / 20, c[-1]), flush=True)
_backend.set_option("row_local_variant", -1)
