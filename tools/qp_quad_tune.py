# scratch: the four-lanes-per-sample QP kernel (qp_mode 4) against lane + wave (qp_mode 2) on the
# benchmark problem: same state in, compare Z / pass counts, time weights_update early and late
import sys, time, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd")); sys.path.insert(0, _R)
import numpy as np
import bench
from convex_dim_red import _backend
n, p, k = bench.N_SAMPLES, bench.N_FEATURES, bench.N_COMPONENTS
if len(sys.argv) > 1: n = int(sys.argv[1])
only_state = sys.argv[2] if len(sys.argv) > 2 else None     # "early" / "late"
only_mode = int(sys.argv[3]) if len(sys.argv) > 3 else None  # profile runs: one state, one mode
X = bench.synthetic_rows(0, n); C0, Z0 = bench.start_factors(n, k)
ctx = _backend.Context(dtype="float32")
ctx.set_data(X)
states = {}
ctx.set_state(C0, Z0, np.ones(k)); ctx.prepare()
ctx.outer_iterations(5, dict(max_iterations=1), {})
states["early (after 5)"] = ctx.get_state()[:2]
ctx.outer_iterations(35, dict(max_iterations=1), {})
states["late (after 40)"] = ctx.get_state()[:2]

def run(label, state, reps=4, **opts):
    for key, val in opts.items(): _backend.set_option(key, val)
    C, Z = states[state]
    ts = []; mx = []; tot = []
    ctx.set_state(C, Z, np.ones(k)); ctx.prepare()
    out = None
    for it in range(reps):
        ctx.dictionary_update(max_iterations=1)
        t = time.perf_counter(); st = ctx.weights_update(); ts.append(time.perf_counter() - t); mx.append(st.max_passes); tot.append(st.total_passes)
        if it == 0: out = ctx.get_state()[1].copy()
    print("%-16s %-34s weights_update ms: %s  max %s mean passes %.1f parked(last) %d" % (state, label, " ".join("%.3f" % (1e3 * t) for t in ts), mx, tot[-1] / n, st.reserved), flush=True)
    return out, tot[0]

if only_mode is not None:
    state = [s for s in states if s.startswith(only_state)][0]
    run("mode %d" % only_mode, state, reps=6, qp_mode=only_mode)
    ctx.close(); sys.exit(0)
for state in states:
    if only_state and not state.startswith(only_state): continue
    ref, tot_ref = run("lane+wave (mode 2)", state, qp_mode=2)
    base = dict(qp_mode=4, qp_quad_cap=24, qp_quad_occ=3, qp_quad_waves=3072)
    for label, opts in (("quad occ 3, 3072 waves", {}),
                        ("quad occ 3, 6250 waves", dict(qp_quad_waves=6250)),
                        ("quad occ 3, 4608 waves", dict(qp_quad_waves=4608)),
                        ("quad occ 4, 6250 waves", dict(qp_quad_occ=4, qp_quad_waves=6250)),
                        ("quad occ 2, 6250 waves", dict(qp_quad_occ=2, qp_quad_waves=6250))):
        o = dict(base); o.update(opts)
        got, tot = run(label, state, reps=7, **o)
ctx.close()
