#!/usr/bin/env python
"""Per outer iteration of the headline problem: QP pass statistics (mean, max, samples parked by
k_qp_quad for k_qp_wave, passes spent in either kernel)."""
import os
import sys

import numpy as np

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd"))
sys.path.insert(0, _R)
import bench  # noqa: E402
from convex_dim_red import _backend  # noqa: E402

n, k = bench.N_SAMPLES, bench.N_COMPONENTS
X = bench.synthetic_rows(0, n)
C0, Z0 = bench.start_factors(n, k)
with _backend.Context(dtype="float32") as ctx:
    ctx.set_data(X)
    ctx.set_state(C0, Z0, np.ones(k))
    ctx.prepare()
    for t in range(60):
        ctx.dictionary_update(max_iterations=1)
        qs = ctx.weights_update()
        it = ctx.qp_iters() if hasattr(ctx, "qp_iters") else None
        line = "t=%2d mean passes %.2f max %d parked %d" % (t, qs.total_passes / float(n), qs.max_passes, qs.reserved)
        if it is not None:
            over = it[it > 24]
            line += "  passes beyond 24: %d (%.1f %% of all), parked histogram >48: %d >96: %d >192: %d" % (
                (over - 24).sum(), 100.0 * (over - 24).sum() / it.sum(), (it > 48).sum(), (it > 96).sum(), (it > 192).sum())
        print(line, flush=True)
