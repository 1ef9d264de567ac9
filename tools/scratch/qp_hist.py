# scratch: per-sample SPG pass counts of the weights QP on the benchmark problem (outer iteration 30)
import sys, time, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd")); sys.path.insert(0, _R)
import numpy as np
import bench
from convex_dim_red import _backend
n, p, k = bench.N_SAMPLES, bench.N_FEATURES, bench.N_COMPONENTS
X = bench.synthetic_rows(0, n); C0, Z0 = bench.start_factors(n, k)
ctx = _backend.Context(dtype="float32")
ctx.set_data(X)
ctx.set_state(C0, Z0, np.ones(k)); ctx.prepare()
for stage in (8, 30, 60):
    ctx.outer_iterations({8: 8, 30: 21, 60: 29}[stage], dict(max_iterations=1), {})
    ctx.dictionary_update(max_iterations=1)
    C, Z, _ = ctx.get_state()
    ZtZ, CKCt, CKZ, tr = ctx.grams()
    CX = ctx.archetypes()
    B = CX.dot(X.astype(np.float64).T)          # k x n
    Zn, it = _backend.qp_batch(CKCt, B, Z, "kn", return_iters=True)
    it = np.asarray(it)
    print("stage %d: mean %.2f max %d; histogram (passes: count)" % (stage, it.mean(), it.max()))
    h = np.bincount(np.minimum(it, 60))
    print(" ".join("%d:%d" % (i, c) for i, c in enumerate(h) if c))
    np.save(os.path.join(_R, "gpurun_out", "qp_iters_stage%d.npy" % stage), it.astype(np.int32))
    ctx.weights_update()
ctx.close()
