# scratch: QP schedule tuning on the benchmark problem
import sys, time
sys.path.insert(0, "matrix-factorization-case-studies_amd"); sys.path.insert(0, ".")
import numpy as np
import bench
from convex_dim_red import _backend
n, p, k = bench.N_SAMPLES, bench.N_FEATURES, bench.N_COMPONENTS
X = bench.synthetic_rows(0, n); C0, Z0 = bench.start_factors(n, k)
ctx = _backend.Context(dtype="float32")
ctx.set_data(X)
def run(label, n_outer=12, **opts):
    for key, val in opts.items(): _backend.set_option(key, val)
    ctx.set_state(C0, Z0, np.ones(k)); ctx.prepare()
    ctx.outer_iterations(2, dict(max_iterations=1), {})
    t = time.perf_counter(); costs = ctx.outer_iterations(n_outer, dict(max_iterations=1), {}); dt = time.perf_counter() - t
    tq = []
    for _ in range(3):
        t = time.perf_counter(); st = ctx.weights_update(); tq.append(time.perf_counter() - t)
    print("%-44s %.3f ms/outer (iters 3-%d); weights_update %.3f ms; final cost %.9f; long+overflow %d max %d"
          % (label, 1e3 * dt / n_outer, 2 + n_outer, 1e3 * min(tq), costs[-1], st.reserved, st.max_passes), flush=True)
run("schedule off", qp_schedule=0)
run("schedule on  long 32 side 64", qp_schedule=1, qp_long_threshold=32, qp_side_blocks=64)
run("schedule on  long 24 side 64", qp_long_threshold=24)
run("schedule on  long 48 side 32", qp_long_threshold=48, qp_side_blocks=32)
run("schedule on  long 32 side 128", qp_long_threshold=32, qp_side_blocks=128)
run("schedule on  long 32 side 64 cap 12", qp_side_blocks=64, qp_pass_cap=12)
run("schedule on  long 32 side 64 cap 24", qp_pass_cap=24)
ctx.close()
