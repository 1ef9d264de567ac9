# scratch: odd problem shapes through the whole device path against the oracle
import sys, os, warnings
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd")); sys.path.insert(0, _R)
import numpy as np
from convex_dim_red import _backend
from oracle import aa_oracle as orc
bad = 0
for (n, p, k) in ((5, 3, 1), (7, 1, 2), (64, 128, 1), (65, 129, 2), (129, 5, 3), (1000, 1, 4), (33, 700, 31), (200, 40, 33), (130, 260, 64), (4097, 3, 5)):
    for dtype in ("float64", "float32"):
        rng = np.random.RandomState(n + p + k)
        X = rng.standard_normal((n, p))
        Xd = X.astype(np.float32).astype(np.float64) if dtype == "float32" else X
        C = orc.right_stochastic_matrix((k, n), rng)
        Z = orc.right_stochastic_matrix((n, k), rng)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            w = orc.iterate_aa(Xd, Z.copy(), C.copy(), np.ones(k), tolerance=0, max_iterations=2,
                               dictionary_solver_kwargs=dict(max_iterations=1), require_monotonic_cost_decrease=False)
        try:
            with _backend.Context(dtype=dtype) as ctx:
                ctx.set_data(X.astype(np.float32) if dtype == "float32" else X)
                ctx.set_state(C, Z, np.ones(k)); ctx.prepare()
                costs = ctx.outer_iterations(2, dict(max_iterations=1), {})
                Cf, Zf, _ = ctx.get_state()
            rel = abs(costs[-1] - w[3]) / max(abs(w[3]), 1e-300)
            ok = rel < (1e-8 if dtype == "float64" else 2e-4) and abs(Zf.sum(axis=1) - 1).max() < 1e-12 and Cf.min() >= 0
            print("n=%5d p=%4d k=%2d %s: cost %.10e oracle %.10e rel %.1e  %s" % (n, p, k, dtype, costs[-1], w[3], rel, "ok" if ok else "MISMATCH"), flush=True)
            bad += not ok
        except Exception as e:
            print("n=%5d p=%4d k=%2d %s: EXCEPTION %s" % (n, p, k, dtype, e), flush=True); bad += 1
print("bad =", bad)
