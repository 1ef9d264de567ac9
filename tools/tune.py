# scratch: QP knob tuning on the benchmark problem
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
ctx.outer_iterations(8, dict(max_iterations=1), {})
C8, Z8, _ = ctx.get_state()
def run(label, **opts):
    for key, val in opts.items(): _backend.set_option(key, val)
    ts = []; mx = []
    ctx.set_state(C8, Z8, np.ones(k)); ctx.prepare()
    for it in range(6):
        ctx.dictionary_update(max_iterations=1)
        t = time.perf_counter(); st = ctx.weights_update(); ts.append(time.perf_counter() - t); mx.append(st.max_passes)
    print("%-40s weights_update ms: %s  max passes %s overflow(last) %d" % (label, " ".join("%.2f" % (1e3 * t) for t in ts), mx, st.reserved), flush=True)
run("cap 24 refill 24")
run("cap 16 refill 24", qp_pass_cap=16)
run("cap 12 refill 24", qp_pass_cap=12)
run("cap 8 refill 24", qp_pass_cap=8)
run("cap 32 refill 24", qp_pass_cap=32)
run("cap 16 refill 16", qp_pass_cap=16, qp_refill_min=16)
run("cap 16 refill 32", qp_pass_cap=16, qp_refill_min=32)
run("cap 24 refill 24 again", qp_pass_cap=24, qp_refill_min=24)
run("wave only", qp_mode=1)
ctx.close()
