# scratch: the weights QP on the benchmark problem around outer iteration 30 (run under rocprofv3)
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
ctx.outer_iterations(int(os.environ.get("QP_AT", "30")), dict(max_iterations=1), {})
print('stream probe: %.4f ms -> %.2f TB/s' % (ctx.time_kernel(2, 20), n * p * 4 / ctx.time_kernel(2, 20) / 1e9), flush=True)
if os.environ.get('QP_PROFILE'): _backend.set_option('qp_profile', 1)
for it in range(6):
    t0 = time.perf_counter(); ctx.dictionary_update(max_iterations=1); t1 = time.perf_counter()
    st = ctx.weights_update(); t2 = time.perf_counter()
    print("dict %.3f ms  weights %.3f ms  mean passes %.2f max %d overflow %d" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), st.total_passes / n, st.max_passes, st.reserved), flush=True)
ctx.close()
