#!/usr/bin/env python
"""Is the point where some float32 runs of bench.py's parity_converged problem end (58.7192) a
local optimum of the exact problem, or a point float32 arithmetic cannot leave?  Run float32 with
qp_quad_cap = 1000 (one of the variants that ends there), then continue from that state in
float64 for 500 iterations; also continue the GOOD float32 end state in float32.  The bad state
is saved for a continuation with the CPU oracle."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "matrix-factorization-case-studies_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from convex_dim_red import _backend  # noqa: E402

N, K = 1500, 32
warnings.simplefilter("ignore")
X = bench.synthetic_rows(0, N).astype(np.float64)
X32 = X.astype(np.float32)
C0, Z0 = bench.start_factors(N, K)
dkw = dict(max_iterations=1)


def rec(Z, C):
    return 0.5 * np.linalg.norm(X - Z.dot(C.dot(X))) ** 2 / N


_backend.set_option("qp_quad_cap", 1000)
with _backend.Context(dtype="float32") as ctx:
    ctx.set_data(X32)
    ctx.set_state(C0, Z0, np.ones(K))
    ctx.prepare()
    costs = ctx.outer_iterations(250, dkw, {})
    Cb, Zb, _ = ctx.get_state()
_backend.set_option("qp_quad_cap", 0)
print("float32 qp_quad_cap=1000: %.6f after 250; curve every 10 from 60: %s" %
      (rec(Zb, Cb), " ".join("%.4f" % c for c in costs[1::2][60:140:5])), flush=True)
np.savez(os.path.join(ROOT, "gpurun_out", "bad_state.npz"), C=Cb, Z=Zb)
for dtype in ("float64", "float32"):
    with _backend.Context(dtype=dtype) as ctx:
        ctx.set_data(X32 if dtype == "float32" else X)
        ctx.set_state(Cb, Zb, np.ones(K))
        ctx.prepare()
        costs = ctx.outer_iterations(500, dkw, {})
        C2, Z2, _ = ctx.get_state()
    print("continued in %s for 500 iterations: %.6f (costs at +1, +10, +100, +500: %s)" %
          (dtype, rec(Z2, C2), " ".join("%.6f" % costs[2 * i - 1] for i in (1, 10, 100, 500))), flush=True)
