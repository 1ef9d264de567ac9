#!/usr/bin/env python
"""Is the LDS-DMA row-local kernel bound by the busiest SIMD?  Back-to-back kernel time at row counts
that give every SIMD the same number of 32-row tiles (98 304 = 256 CUs x 12 waves x 32) against the
headline's 100 000 (13 waves on a CU: one SIMD has four), for several waves-per-block settings;
the reduce-over-rows kernel beside it."""
import os
import sys

import numpy as np

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd"))
sys.path.insert(0, _R)
import bench  # noqa: E402
from convex_dim_red import _backend  # noqa: E402

k = bench.N_COMPONENTS
for n in (98304, 100000, 131072):
    X = bench.synthetic_rows(0, n)
    C0, Z0 = bench.start_factors(n, k)
    with _backend.Context(dtype="float32") as ctx:
        ctx.set_data(X)
        ctx.set_state(C0, Z0, np.ones(k))
        ctx.prepare()
        for W in (0, 12, 13, 16):
            _backend.set_option("row_local_waves", W)
            t = ctx.time_kernel(1, 20)
            print("n=%d row_local_waves=%d: row-local %.4f ms (%.2f TB/s), %.3f us per 1000 rows"
                  % (n, W, t, n * 4096 * 4 / t / 1e9, 1e3 * t / (n / 1000.0)), flush=True)
        _backend.set_option("row_local_waves", 0)
        t = ctx.time_kernel(0, 20)
        print("n=%d reduce-rows %.4f ms (%.2f TB/s)" % (n, t, n * 4096 * 4 / t / 1e9), flush=True)
