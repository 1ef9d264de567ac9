# scratch: how well does a sample's pass count in one weights update predict the next one's?
# (decides whether predicted-long samples can be started early in the low-latency kernel)
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd")); sys.path.insert(0, _R)
import numpy as np
import bench
from convex_dim_red import _backend
n, p, k = 100000, bench.N_FEATURES, bench.N_COMPONENTS
X = bench.synthetic_rows(0, n).astype(np.float64); C0, Z0 = bench.start_factors(n, k)
ctx = _backend.Context(dtype="float64")
ctx.set_data(X)
ctx.set_state(C0, Z0, np.ones(k)); ctx.prepare()
for stage, skip in (("early", 5), ("late", 30)):
    ctx.outer_iterations(skip, dict(max_iterations=1), {})
    its = []
    for t in range(3):
        ctx.dictionary_update(max_iterations=1)
        C, Z, _ = ctx.get_state()
        CX = C.dot(X)
        A, B = CX.dot(CX.T), CX.dot(X.T)
        _, it = _backend.qp_batch(A, B, Z, "kn", return_iters=True)
        its.append(it)
        ctx.weights_update()
    for a, b in ((its[0], its[1]), (its[1], its[2])):
        top = np.argsort(-b)[:20]
        print(stage, "top-20 now:", b[top].tolist())
        print(stage, "  their previous counts:", a[top].tolist())
        for thr in (100, 200):
            long_now = b >= thr
            if long_now.sum():
                print(stage, "  samples with >= %d passes now: %d; of these had < 24 before: %d, < 48: %d, < 64: %d" % (
                    thr, long_now.sum(), (a[long_now] < 24).sum(), (a[long_now] < 48).sum(), (a[long_now] < 64).sum()))
        print(stage, "  corr %.3f; samples >= 48 before: %d, >= 64: %d, >= 24: %d" % (np.corrcoef(a, b)[0, 1], (a >= 48).sum(), (a >= 64).sum(), (a >= 24).sum()), flush=True)
ctx.close()
