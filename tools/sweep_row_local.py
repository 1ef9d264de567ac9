# scratch: time the row-local GEMM variants on the benchmark shard (HIP events, aa_time_kernel)
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd")); sys.path.insert(0, _R)
import numpy as np
import bench
from convex_dim_red import _backend
n, p, k = bench.N_SAMPLES, bench.N_FEATURES, bench.N_COMPONENTS
X = bench.synthetic_rows(0, n); C0, Z0 = bench.start_factors(n, k)
ctx = _backend.Context(dtype="float32")
ctx.set_data(X); ctx.set_state(C0, Z0, np.ones(k)); ctx.prepare()
ref = None
for v, w, sg in ((8, 0, 0), (4, 0, 0), (8, 0, 0)):
    _backend.set_option("row_local_variant", v); _backend.set_option("row_local_waves", w); _backend.set_option("row_local_stagger", sg)
    ctx.time_kernel(1, 5)
    ms = ctx.time_kernel(1, 40)
    ctx.set_state(C0, Z0, np.ones(k)); ctx.prepare()
    g = ctx.grams()[2]
    if ref is None: ref = g
    print("variant %d waves %d stagger %d: %.4f ms  %.2f TB/s   max|dCKZ| %.2e" % (v, w, sg, ms, n * p * 4 / ms / 1e9, np.abs(g - ref).max()), flush=True)
print("reduce_rows: %.4f ms" % ctx.time_kernel(0, 40))
for w, label in ((2, "4 loads x 4096 blocks"), (5, "8 x 8192"), (6, "ws pattern, row-major"), (7, "ws pattern, tiled 8 KB"), (6, "ws pattern, row-major"), (7, "ws pattern, tiled 8 KB")):
    ctx.time_kernel(w, 3); ms = ctx.time_kernel(w, 20)
    print("stream probe %-22s %.4f ms  %.2f TB/s" % (label, ms, n * p * 4 / ms / 1e9), flush=True)
_backend.set_option("row_local_variant", -1); _backend.set_option("row_local_waves", 0); _backend.set_option("row_local_stagger", 0)
ctx.close()
