# scratch: the benchmark iteration through the multi-rank code path on ONE GPU (1-rank RCCL
# communicator, AA_FORCE_RCCL=1): what the collectives' launches and the projection's host
# checks cost before any real network latency
import sys, os, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd")); sys.path.insert(0, _R)
import numpy as np
import bench
from convex_dim_red import _backend
n = int(sys.argv[1]) if len(sys.argv) > 1 else bench.N_SAMPLES
p, k = bench.N_FEATURES, bench.N_COMPONENTS
X = bench.synthetic_rows(0, n, n, p, k); C0, Z0 = bench.start_factors(n, k)
for force in (False, True, "p2p"):
    if force is True: os.environ["AA_FORCE_RCCL"] = "1"
    else: os.environ.pop("AA_FORCE_RCCL", None)
    ctx = _backend.Context(dtype="float32")
    if force is True: ctx.comm_init(_backend.comm_unique_id(), 0, 1)
    if force == "p2p": ctx.p2p_init(0, 1, "forced")
    ctx.set_data(X); ctx.set_state(C0, Z0, np.ones(k)); ctx.prepare()
    ctx.outer_iterations(5, dict(max_iterations=1), {})
    t = time.perf_counter(); costs = ctx.outer_iterations(50, dict(max_iterations=1), {}); t = time.perf_counter() - t
    print("n=%d %s: %.3f ms per outer iteration, final cost %.9f" % (n, {False: "direct path", True: "multi-rank path (1-rank RCCL)", "p2p": "multi-rank path (1-rank peer-to-peer all-reduce)"}[force], 1e3 * t / 50, costs[-1]), flush=True)
    ctx.close()
