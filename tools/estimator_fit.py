# scratch: wall time of a fit through the estimator API on the benchmark data
import sys, os, time, warnings
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd")); sys.path.insert(0, _R)
import numpy as np
import bench, convex_dim_red as cdr
n, p, k = bench.N_SAMPLES, bench.N_FEATURES, bench.N_COMPONENTS
X = bench.synthetic_rows(0, n)
for dtype in ("float32", "float64"):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = cdr.ArchetypalAnalysis(k, init="random", random_state=0, tolerance=0.0, max_iterations=100,
                                   dictionary_solver_kwargs=dict(max_iterations=1), dtype=dtype)
        t = time.perf_counter(); W = m.fit_transform(X if dtype == "float32" else X.astype(np.float64)); t = time.perf_counter() - t
    print("%s: fit_transform 100 iterations in %.2f s (%.2f ms per iteration reported by the estimator), cost %.6f, n_iter %d"
          % (dtype, t, 1e3 * m.avg_time_per_iter, m.cost, m.n_iter), flush=True)
