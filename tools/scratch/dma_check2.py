#!/usr/bin/env python
"""Headline problem, 25 outer iterations from bench.py's start: cost with the register-staged
wave-streaming row-local kernel (one fp32 chain per row), with the LDS-DMA kernel (float64 sums of
32-column pieces) and in float64 data -- which float32 variant follows the float64 run?"""
import os
import sys
import time

import numpy as np

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd"))
sys.path.insert(0, _R)
import bench  # noqa: E402
from convex_dim_red import _backend  # noqa: E402

n, p, k = bench.N_SAMPLES, bench.N_FEATURES, bench.N_COMPONENTS
X = bench.synthetic_rows(0, n)
C0, Z0 = bench.start_factors(n, k)
dkw = dict(max_iterations=1)
res = {}
with _backend.Context(dtype="float32") as ctx:
    ctx.set_data(X)
    for tag, opts in (("ws (variant 8)", dict(row_local_variant=8)),
                      ("dma ring 8", dict(row_local_variant=9, row_local_ring=8)),
                      ("dma ring 8 nt", dict(row_local_variant=9, row_local_ring=8, row_local_nt=1)),
                      ("dma ring 11", dict(row_local_variant=9, row_local_ring=11)),
                      ("ws (variant 8)", dict(row_local_variant=8)),
                      ("dma ring 8", dict(row_local_variant=9, row_local_ring=8))):
        for name, v in opts.items():
            _backend.set_option(name, v)
        ctx.set_state(C0, Z0, np.ones(k))
        ctx.prepare()
        c5 = ctx.outer_iterations(5, dkw, {})
        t0 = time.perf_counter()
        c = ctx.outer_iterations(20, dkw, {})
        dt = time.perf_counter() - t0
        c2 = ctx.outer_iterations(30, dkw, {})
        res[tag] = np.concatenate([c5, c, c2])[1::2]
        print("%-16s %.3f ms per outer iteration (5..25)" % (tag, 1e3 * dt / 20), flush=True)
        for name in opts:
            _backend.set_option(name, {"row_local_variant": -1}.get(name, 0))
Xd = X.astype(np.float64)
with _backend.Context(dtype="float64") as ctx:
    ctx.set_data(Xd)
    ctx.set_state(C0, Z0, np.ones(k))
    ctx.prepare()
    ref = ctx.outer_iterations(55, dkw, {})[1::2]
for tag, c in res.items():
    rel = np.abs(c - ref) / ref
    print("%-16s |cost - float64| / cost after 1, 5, 10, 25, 55 iterations: %s" %
          (tag, " ".join("%.2e" % rel[i] for i in (0, 4, 9, 24, 54))), flush=True)
