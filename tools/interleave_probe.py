#!/usr/bin/env python
"""Feasibility: R contexts of one device that share the data matrix, driven by ONE host thread:
batches of 8 outer iterations enqueued round-robin without synchronising, against the same work
done context after context."""
import os
import sys
import time

import numpy as np

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd"))
sys.path.insert(0, _R)
from convex_dim_red import _backend  # noqa: E402
from oracle import aa_oracle as orc  # noqa: E402

n, p, k = 1610, 25000, 5
rng = np.random.RandomState(0)
B = rng.standard_normal((k, p))
Zt = orc.right_stochastic_matrix((n, k), rng) ** 4
Zt /= Zt.sum(axis=1, keepdims=True)
X = (Zt.dot(B) + 0.05 * rng.standard_normal((n, p))).astype(np.float32)
dkw = dict(max_iterations=1)
T = 96
for R in (1, 2, 4, 8):
    ctxs = []
    owner = _backend.Context(dtype="float32")
    owner.set_data(X)
    ctxs.append(owner)
    for r in range(1, R):
        c = _backend.Context(dtype="float32")
        c.share_data(owner)
        ctxs.append(c)
    starts = []
    for r in range(R):
        rs = np.random.RandomState(10 + r)
        starts.append((orc.right_stochastic_matrix((k, n), rs), orc.right_stochastic_matrix((n, k), rs)))

    def reset():
        for c, (C0, Z0) in zip(ctxs, starts):
            c.set_state(C0, Z0, np.ones(k))
            c.prepare()
            c.outer_iterations(8, dkw, {})

    reset()
    t0 = time.perf_counter()
    for c in ctxs:
        for b in range(T // 8):
            c.outer_iterations(8, dkw, {})
    t_seq = time.perf_counter() - t0
    want = [c.cost() for c in ctxs]
    reset()
    _backend.set_option("outer_nosync", 1)
    t0 = time.perf_counter()
    for b in range(T // 8):
        for c in ctxs:
            c.outer_iterations_nocost(8, dkw, {})
    t_enq = time.perf_counter() - t0
    got = [c.cost() for c in ctxs]              # synchronises every stream
    t_int = time.perf_counter() - t0
    _backend.set_option("outer_nosync", 0)
    print("R=%d: sequential %.1f ms (%.0f it/s), interleaved %.1f ms (%.0f it/s; host enqueue %.1f ms), costs equal: %s"
          % (R, 1e3 * t_seq, R * T / t_seq, 1e3 * t_int, R * T / t_int, 1e3 * t_enq, got == want), flush=True)
    for c in ctxs[1:]:
        c.close()
    owner.close()
