"""Throughput of the two driver configurations (BASELINE.json configs[1], configs[2]) on their
synthetic stand-ins, float64 (the reference dtype), through the loops the estimators run
(aa_iterate / aa_gpnh_iterate with tolerance 0: every iteration includes the device-side
monotonicity check / stopping rule).  One JSON line per configuration: it/s, ms per iteration,
algorithmic HBM bytes per iteration (passes over X) and the GB/s they correspond to.  These
problems are 322 MB / 30 MB: latency-bound, a few dozen dependent launches per iteration."""
import json
import os
import sys
import time
import warnings

import numpy as np

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd"))
sys.path.insert(0, _R)
from convex_dim_red import _backend  # noqa: E402
from oracle import aa_oracle as orc  # noqa: E402

warnings.simplefilter("ignore")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100


def c2():
    n, p, k = 1610, 25000, 5
    rng = np.random.RandomState(0)
    B = rng.standard_normal((k, p))
    Zt = orc.right_stochastic_matrix((n, k), rng) ** 4
    Zt /= Zt.sum(axis=1, keepdims=True)
    X = Zt.dot(B) + 0.05 * rng.standard_normal((n, p))
    rs = np.random.RandomState(1)
    return X, orc.right_stochastic_matrix((k, n), rs), orc.right_stochastic_matrix((n, k), rs), k


def c3():
    n, p, k = 22280, 167, 10
    rng = np.random.RandomState(0)
    W0 = rng.standard_normal((p, k))
    X = orc.right_stochastic_matrix((n, k), rng).dot(W0.T) + 0.1 * rng.standard_normal((n, p))
    rs = np.random.RandomState(1)
    return X, np.sqrt(np.abs(X).mean() / k) * rs.randn(p, k), orc.right_stochastic_matrix((n, k), rs), k


X, C0, Z0, k = c2()
with _backend.Context(dtype="float64") as ctx:
    ctx.set_data(X)
    ctx.set_state(C0, Z0, np.ones(k))
    cost = ctx.prepare()
    costs, st = ctx.iterate(cost, 10, 0.0, "abs_delta_f", False, True, True, dict(max_iterations=1), {})
    t0 = time.perf_counter()
    costs, st = ctx.iterate(costs[-1], steps, 0.0, "abs_delta_f", False, True, True, dict(max_iterations=1), {})
    dt = time.perf_counter() - t0
n, p = X.shape
bytes_it = 4.0 * n * p * 8
print(json.dumps({"config": "C2 stand-in: AA k=5, 1610 x 25000, float64, production solver settings",
                  "it_per_s": steps / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps,
                  "algorithmic_bytes_per_step": bytes_it, "hbm_GBps": bytes_it * steps / dt / 1e9,
                  "cost_last": float(costs[-1])}), flush=True)

X, W0, Z0, k = c3()
n, p = X.shape
for lam in (0.0, 1.0):
    with _backend.Context(dtype="float64") as ctx:
        ctx.set_data(X)
        ctx.gpnh_set_factors(k, W=W0, Z=Z0)
        ctx.gpnh_iterate(lam, 10, 0.0, "rel_delta_f", False, True, True, dict(max_iterations=1))
        t0 = time.perf_counter()
        c0, costs, st = ctx.gpnh_iterate(lam, steps, 0.0, "rel_delta_f", False, True, True, dict(max_iterations=1))
        dt = time.perf_counter() - t0
    bytes_it = 2.0 * n * p * 8
    print(json.dumps({"config": "C3 stand-in: GPNH k=10, 22280 x 167, lambda_W=%g, float64, weights QP max_iterations=1" % lam,
                      "it_per_s": steps / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps,
                      "algorithmic_bytes_per_step": bytes_it, "hbm_GBps": bytes_it * steps / dt / 1e9,
                      "cost_last": float(costs[-1])}), flush=True)

# KernelAA on the implicit linear kernel of the headline problem (the explicit
# 100 000 x 100 000 kernel matrix would be 80 GB); float32 data, the benchmark's solver settings
if os.environ.get("BENCH_CONFIGS_KERNEL", "1") != "0":
    import bench  # noqa: E402
    n, p, k = bench.N_SAMPLES, bench.N_FEATURES, bench.N_COMPONENTS
    X = bench.synthetic_rows(0, n)
    C0, Z0 = bench.start_factors(n, k)
    with _backend.Context(dtype="float32") as ctx:
        ctx.set_data(X)
        ctx.set_linear_kernel(True)
        ctx.set_state(C0, Z0, np.ones(k))
        cost = ctx.prepare()
        costs, st = ctx.iterate(cost, 5, 0.0, "abs_delta_f", False, True, True, dict(max_iterations=1), {})
        ksteps = min(steps, 50)
        t0 = time.perf_counter()
        costs, st = ctx.iterate(costs[-1], ksteps, 0.0, "abs_delta_f", False, True, True, dict(max_iterations=1), {})
        dt = time.perf_counter() - t0
    bytes_it = 4.0 * n * p * 4
    print(json.dumps({"config": "KernelAA on the implicit linear kernel K = XX', X = 100000 x 4096 float32, k=32 (K never formed)",
                      "it_per_s": ksteps / dt, "ms_per_step": 1e3 * dt / ksteps, "steps": ksteps,
                      "algorithmic_bytes_per_step": bytes_it, "hbm_GBps": bytes_it * ksteps / dt / 1e9,
                      "cost_last": float(costs[-1])}), flush=True)
