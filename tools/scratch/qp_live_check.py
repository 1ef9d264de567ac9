#!/usr/bin/env python
"""qp_live (k_qp_quad hands parked samples to a concurrent k_qp_wave launch) against the
two-launch default: per-sample arithmetic does not depend on who continues a sample, so weights
and pass counts must be bit-identical.  Then the cost trace of 40 outer iterations on the
headline problem, both ways."""
import os
import sys

import numpy as np

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd"))
sys.path.insert(0, _R)
import bench  # noqa: E402
from convex_dim_red import _backend  # noqa: E402
from oracle import aa_oracle as orc  # noqa: E402

rng = np.random.RandomState(5)
for n, k, p in ((700, 12, 40), (20000, 32, 64), (100000, 20, 48)):
    W = rng.standard_normal((k, p))
    Zt = orc.right_stochastic_matrix((n, k), rng) ** 3
    Zt /= Zt.sum(axis=1, keepdims=True)
    Xs = Zt.dot(W) + 0.05 * rng.standard_normal((n, p))
    A, B = W.dot(W.T), W.dot(Xs.T)
    Z0 = orc.right_stochastic_matrix((n, k), rng)
    _backend.set_option("qp_mode", 4)
    outs = []
    for live in (0, 1, 1):
        _backend.set_option("qp_live", live)
        outs.append(_backend.qp_batch(A, B, Z0, "kn", return_iters=True))
    _backend.set_option("qp_live", 0)
    _backend.set_option("qp_mode", 0)
    for got, it in outs[1:]:
        assert np.array_equal(got, outs[0][0]) and np.array_equal(it, outs[0][1]), (n, k)
    print("n=%d k=%d: identical (passes mean %.1f max %d, %d samples beyond 24)"
          % (n, k, outs[0][1].mean(), outs[0][1].max(), (outs[0][1] > 24).sum()), flush=True)

n, k = bench.N_SAMPLES, bench.N_COMPONENTS
X = bench.synthetic_rows(0, n)
C0, Z0 = bench.start_factors(n, k)
traces = []
with _backend.Context(dtype="float32") as ctx:
    ctx.set_data(X)
    for live in (0, 1):
        _backend.set_option("qp_live", live)
        ctx.set_state(C0, Z0, np.ones(k))
        ctx.prepare()
        traces.append(np.asarray(ctx.outer_iterations(40, dict(max_iterations=1), {})))
    _backend.set_option("qp_live", 0)
assert np.array_equal(traces[0], traces[1]), np.abs(traces[0] - traces[1]).max()
print("headline, 40 outer iterations: cost traces identical", flush=True)
print("QP_LIVE_OK")
