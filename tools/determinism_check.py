# scratch: run-to-run determinism of the small end-to-end problem of test_rccl_path_single_rank
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd")); sys.path.insert(0, _R)
import numpy as np
from convex_dim_red import _backend
from oracle import aa_oracle as orc
rng = np.random.RandomState(17)
n, p, k = 900, 260, 6
X = rng.standard_normal((n, p)).astype(np.float32)
C = orc.right_stochastic_matrix((k, n), rng)
Z = orc.right_stochastic_matrix((n, k), rng)
def run(force, **opts):
    if force: os.environ["AA_FORCE_RCCL"] = "1"
    for a, b in opts.items(): _backend.set_option(a, b)
    try:
        with _backend.Context(dtype="float32") as ctx:
            if force: ctx.comm_init(_backend.comm_unique_id(), 0, 1)
            ctx.set_data(X); ctx.set_state(C, Z, np.ones(k)); ctx.prepare()
            outs = []
            for it in range(3):
                ctx.dictionary_update(max_iterations=1); c1 = ctx.get_state()[0]
                ctx.weights_update(); z1 = ctx.get_state()[1]
                outs += [c1, z1]
            return outs
    finally:
        os.environ.pop("AA_FORCE_RCCL", None)
        for a in opts: _backend.set_option(a, 0)
def cmp(a, b, label):
    print(label, ["%.1e" % np.abs(x - y).max() for x, y in zip(a, b)])
for mode in (1, 2):
    r = [run(False, qp_mode=mode) for _ in range(3)]
    cmp(r[0], r[1], "direct qp_mode=%d run0 vs run1" % mode); cmp(r[0], r[2], "direct qp_mode=%d run0 vs run2" % mode)
    f = [run(True, qp_mode=mode) for _ in range(2)]
    cmp(f[0], f[1], "rccl   qp_mode=%d run0 vs run1" % mode); cmp(r[0], f[0], "direct vs rccl qp_mode=%d" % mode)
