# scratch: per-pass latency of the two QP kernels
import sys, time
import os; _R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd")); sys.path.insert(0, _R)
import numpy as np
from convex_dim_red import _backend
from oracle import aa_oracle as orc
def problem(n, k, seed=1):
    rng = np.random.RandomState(seed); p = 2 * k + 5
    W = rng.standard_normal((k, p))
    Zt = orc.right_stochastic_matrix((n, k), rng) ** 3; Zt /= Zt.sum(axis=1, keepdims=True)
    Xs = Zt.dot(W) + 0.05 * rng.standard_normal((n, p))
    return W.dot(W.T), W.dot(Xs.T), orc.right_stochastic_matrix((n, k), rng)
def timed(A, B, Z0, **kw):
    _backend.qp_batch(A, B, Z0, "kn", **kw)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); Z, it = _backend.qp_batch(A, B, Z0, "kn", return_iters=True, **kw); best = min(best, time.perf_counter() - t)
    return best, it
k = 32
A, B, Z0 = problem(64, k)
t0, _ = timed(A, B, Z0, max_iterations=1, epsilon_two=0.0, epsilon_one=0.0)
for cap, label in ((1, "wave kernel"), (100000, "lane kernel"), (-1, "row kernel")):
    _backend.set_option("qp_mode", 1 if cap == 1 else (3 if cap < 0 else 2))
    _backend.set_option("qp_pass_cap", max(cap, 1))
    for iters in (50, 200):
        t, it = timed(A, B, Z0, max_iterations=iters, epsilon_two=0.0, epsilon_one=0.0, max_feval=10**8)
        print("%s: n=64, %d forced passes: %.3f ms total, %.2f us per pass (call overhead ~%.3f ms)"
              % (label, iters, 1e3 * t, 1e6 * (t - t0) / (iters - 1), 1e3 * t0), flush=True)
A, B, Z0 = problem(65536, k)
_backend.set_option("qp_mode", 2)
_backend.set_option("qp_pass_cap", 100000)
t1, _ = timed(A, B, Z0, max_iterations=1, epsilon_two=0.0, epsilon_one=0.0)
t, it = timed(A, B, Z0, max_iterations=17, epsilon_two=0.0, epsilon_one=0.0, max_feval=10**8)
print("lane kernel: n=65536 (1024 waves), 16 extra forced passes: %.3f ms -> %.2f us per pass"
      % (1e3 * (t - t1), 1e6 * (t - t1) / 16), flush=True)

_backend.set_option("qp_mode", 3)
_backend.set_option("qp_profile", 1)
for nn in (4, 8192, 65536):
    A, B, Z0 = problem(nn, k)
    t1, _ = timed(A, B, Z0, max_iterations=1, epsilon_two=0.0, epsilon_one=0.0)
    t, it = timed(A, B, Z0, max_iterations=17, epsilon_two=0.0, epsilon_one=0.0, max_feval=10**8)
    print("row kernel: n=%d, 16 extra forced passes: %.3f ms -> %.2f us per pass" % (nn, 1e3 * (t - t1), 1e6 * (t - t1) / 16), flush=True)
    t, it = timed(A, B, Z0)
    print("row kernel: n=%d to convergence: %.3f ms, mean passes %.2f max %d" % (nn, 1e3 * t, it.mean(), it.max()), flush=True)
