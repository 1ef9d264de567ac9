# scratch: time the reduce-over-rows GEMM for different unroll depths / block counts
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd")); sys.path.insert(0, _R)
import numpy as np
import bench
from convex_dim_red import _backend
n, p, k = bench.N_SAMPLES, bench.N_FEATURES, bench.N_COMPONENTS
X = bench.synthetic_rows(0, n); C0, Z0 = bench.start_factors(n, k)
ref = None
for u, nb in ((4, 512), (8, 512), (4, 768), (8, 768), (4, 1024), (8, 1024), (4, 384), (8, 256), (4, 512)):
    _backend.set_option("reduce_rows_unroll", u); _backend.set_option("reduce_rows_blocks", nb)
    ctx = _backend.Context(dtype="float32")
    ctx.set_data(X); ctx.set_state(C0, Z0, np.ones(k)); ctx.prepare()
    g = ctx.grams()[2]
    if ref is None: ref = g
    ctx.time_kernel(0, 5)
    ms = ctx.time_kernel(0, 40)
    print("unroll %d blocks %d: %.4f ms  %.2f TB/s  max|dCKZ| %.2e" % (u, nb, ms, n * p * 4 / ms / 1e9, np.abs(g - ref).max()), flush=True)
    ctx.close()
_backend.set_option("reduce_rows_unroll", 4); _backend.set_option("reduce_rows_blocks", 512)
