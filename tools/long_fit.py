# scratch: a long fit through the estimator API on the benchmark data (device loop, float32):
# cost trajectory monotone within the float32 noise band, stopping rule fires, wall time
import sys, os, time, warnings
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd")); sys.path.insert(0, _R)
import numpy as np
import bench, convex_dim_red as cdr
n, p, k = bench.N_SAMPLES, bench.N_FEATURES, bench.N_COMPONENTS
X = bench.synthetic_rows(0, n)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m = cdr.ArchetypalAnalysis(k, init="random", random_state=0, tolerance=1e-7, max_iterations=iters,
                               dictionary_solver_kwargs=dict(max_iterations=1), dtype="float32")
    t = time.perf_counter(); W = m.fit_transform(X); t = time.perf_counter() - t
d = np.asarray(m.cost_deltas)
print("float32: %d outer iterations in %.2f s (%.3f ms each incl. upload %.2f s total), cost %.8f, largest cost increase %.2e, "
      "weights on the simplex: %s" % (m.n_iter + 1, t, 1e3 * m.avg_time_per_iter, t, m.cost, max(0.0, float(d.max())),
                                        bool(np.all(W >= 0) and np.allclose(W.sum(axis=1), 1, atol=1e-12))), flush=True)
