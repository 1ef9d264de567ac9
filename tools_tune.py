# scratch: on-GPU tuning sweep (row-local variants, QP pass cap); not part of the product
import sys, time, json
sys.path.insert(0, "matrix-factorization-case-studies_amd"); sys.path.insert(0, ".")
import numpy as np
import bench
from convex_dim_red import _backend
n, p, k = bench.N_SAMPLES, bench.N_FEATURES, bench.N_COMPONENTS
t = time.time(); X = bench.synthetic_rows(0, n); C0, Z0 = bench.start_factors(n, k)
print("datagen %.1fs" % (time.time() - t), flush=True)
ctx = _backend.Context(dtype="float32")
ctx.set_data(X); ctx.set_state(C0, Z0, np.ones(k)); ctx.prepare()
ctx.outer_iterations(3, dict(max_iterations=1), {})
C3, Z3, _ = ctx.get_state()
print("reduce_rows ms:", ctx.time_kernel(0, 10), flush=True)
for v in (0, 1, 2):
    _backend.set_option("row_local_variant", v)
    print("row_local variant %d ms: %.4f" % (v, ctx.time_kernel(1, 10)), flush=True)
_backend.set_option("row_local_variant", 2)
for cap in (8, 12, 16):
    _backend.set_option("qp_pass_cap", cap)
    ts = []
    for rep in range(2):
        ctx.set_state(C3, Z3, np.ones(k)); ctx.prepare()
        t = time.perf_counter(); st = ctx.weights_update(); ts.append(time.perf_counter() - t)
    print("qp cap %4d: weights_update %.3f ms (min of 2)  passes total %d max %d overflow %d"
          % (cap, 1e3 * min(ts), st.total_passes, st.max_passes, st.reserved), flush=True)
_backend.set_option("qp_pass_cap", 16)
_backend.set_option("qp_mode", 1)
for rep in range(2):
    ctx.set_state(C3, Z3, np.ones(k)); ctx.prepare()
    t = time.perf_counter(); st = ctx.weights_update(); dt = time.perf_counter() - t
    print("qp wave-only: weights_update %.3f ms  passes total %d max %d" % (1e3 * dt, st.total_passes, st.max_passes), flush=True)
# early state (heavy tail): right after prepare from the random start
for mode in (0, 1):
    _backend.set_option("qp_mode", mode)
    ctx.set_state(C0, Z0, np.ones(k)); ctx.prepare(); ctx.dictionary_update(max_iterations=1)
    t = time.perf_counter(); st = ctx.weights_update(); dt = time.perf_counter() - t
    print("[iter 1] qp mode %d: weights_update %.3f ms  passes total %d max %d overflow %d" % (mode, 1e3 * dt, st.total_passes, st.max_passes, st.reserved), flush=True)
_backend.set_option("qp_mode", 0)
# later-iteration state (QP work changes as the factors converge)
ctx.set_state(C3, Z3, np.ones(k)); ctx.prepare(); ctx.outer_iterations(25, dict(max_iterations=1), {})
C9, Z9, _ = ctx.get_state()
for cap in (12,):
    _backend.set_option("qp_pass_cap", cap)
    ctx.set_state(C9, Z9, np.ones(k)); ctx.prepare()
    t = time.perf_counter(); st = ctx.weights_update(); dt = time.perf_counter() - t
    print("[iter 28] qp cap %4d: weights_update %.3f ms  passes total %d max %d overflow %d"
          % (cap, 1e3 * dt, st.total_passes, st.max_passes, st.reserved), flush=True)
_backend.set_option("qp_mode", 1)
ctx.set_state(C9, Z9, np.ones(k)); ctx.prepare()
t = time.perf_counter(); st = ctx.weights_update(); dt = time.perf_counter() - t
print("[iter 28] qp wave-only: weights_update %.3f ms  passes total %d max %d" % (1e3 * dt, st.total_passes, st.max_passes), flush=True)
_backend.set_option("qp_mode", 0)
t = time.perf_counter(); st = ctx.dictionary_update(max_iterations=1); print("dictionary_update %.3f ms" % (1e3 * (time.perf_counter() - t)))
t = time.perf_counter(); st = ctx.dictionary_update(max_iterations=1); print("dictionary_update %.3f ms (warm)" % (1e3 * (time.perf_counter() - t)))
ctx.close()
