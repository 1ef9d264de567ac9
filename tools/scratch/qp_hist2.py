# scratch: per-sample pass counts of two consecutive weights updates (is the previous count a
# useful predictor for batching samples of similar length?)
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd")); sys.path.insert(0, _R)
import numpy as np
import bench
from convex_dim_red import _backend
n, p, k = bench.N_SAMPLES, bench.N_FEATURES, bench.N_COMPONENTS
X = bench.synthetic_rows(0, n); C0, Z0 = bench.start_factors(n, k)
Xd = X.astype(np.float64)
ctx = _backend.Context(dtype="float32")
ctx.set_data(X); ctx.set_state(C0, Z0, np.ones(k)); ctx.prepare()
ctx.outer_iterations(40, dict(max_iterations=1), {})
its = []
for rep in range(3):
    ctx.dictionary_update(max_iterations=1)
    C, Z, _ = ctx.get_state()
    CKCt = ctx.grams()[1]
    B = ctx.archetypes().dot(Xd.T)
    Zn, it = _backend.qp_batch(CKCt, B, Z, "kn", return_iters=True)
    its.append(np.asarray(it, dtype=np.int32))
    ctx.weights_update()
its = np.stack(its)
np.save(os.path.join(_R, "gpurun_out", "qp_iters_consecutive.npy"), its)
print("corr(t, t+1) = %.3f, corr(t+1, t+2) = %.3f" % (np.corrcoef(its[0], its[1])[0, 1], np.corrcoef(its[1], its[2])[0, 1]))
cap = 24
cur = np.minimum(its[1], cap)
def trips(order):
    b = cur[order][: (n // 64) * 64].reshape(-1, 64)
    return b.max(axis=1)
nat = trips(np.arange(n))
srt = trips(np.argsort(its[0], kind="stable"))
orc = trips(np.argsort(its[1], kind="stable"))
print("mean trips per 64-sample batch: natural order %.2f, sorted by previous count %.2f, sorted by true count %.2f" % (nat.mean(), srt.mean(), orc.mean()))
ctx.close()
