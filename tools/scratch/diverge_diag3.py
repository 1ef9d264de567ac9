#!/usr/bin/env python
"""Third round: what happens inside the float32 / column-split run around the iteration where it
stays behind (t ~ 78..90 on bench.py's parity_converged problem)?  Host-driven loop (one
dictionary update, one weights update at a time -- bit-identical to the device loop), per
iteration: SPG statistics of the dictionary update (f, function evaluations = back-tracking,
flags, residual norm), costs after both updates, QP pass statistics, arg-max changes of the
dictionary.  Families: float32 split on, float32 split off, float64.
Writes gpurun_out/diverge_diag3.log."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "matrix-factorization-case-studies_amd"))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from convex_dim_red import _backend  # noqa: E402

N, K, T = 1500, 32, int(os.environ.get("DIAG_T", "130"))
LOG = open(os.path.join(ROOT, "gpurun_out", "diverge_diag3.log"), "w")


def say(s):
    print(s, flush=True)
    LOG.write(s + "\n")
    LOG.flush()


def main():
    warnings.simplefilter("ignore")
    X = bench.synthetic_rows(0, N).astype(np.float64)
    X32 = X.astype(np.float32)
    C0, Z0 = bench.start_factors(N, K)
    dkw = dict(max_iterations=1)
    fams = [("f32 split=1", "float32", 1), ("f32 split=0", "float32", 0), ("f64", "float64", 1)]
    rows = {}
    for tag, dtype, split in fams:
        _backend.set_option("row_local_split", split)
        out = []
        with _backend.Context(dtype=dtype) as ctx:
            ctx.set_data(X32 if dtype == "float32" else X)
            ctx.set_state(C0, Z0, np.ones(K))
            ctx.prepare()
            prev_arg = C0.argmax(axis=1)
            for t in range(T):
                st = ctx.dictionary_update(**dkw)
                sc = ctx.spg_scalars()
                c_d = ctx.cost()
                qs = ctx.weights_update()
                c_w = ctx.cost()
                rc = ctx.reconstruction_cost()
                C, Z, _ = ctx.get_state()
                arg = C.argmax(axis=1)
                moved = np.nonzero(arg != prev_arg)[0]
                prev_arg = arg
                out.append(dict(t=t, f=st.f, nfe=st.n_feval, flags=st.flags, res=st.res_norm, c_d=c_d, c_w=c_w, rc=rc,
                                qp_mean=qs.total_passes / float(N), qp_max=qs.max_passes, parked=qs.reserved,
                                sc=sc, moved=moved.tolist(), cmax=float(C.max(axis=1).min()),
                                nnzC=int((C > 0).sum()), nnzZ=int((Z > 0).sum())))
        rows[tag] = out
    _backend.set_option("row_local_split", 1)
    for t in range(T):
        if not (8 <= t <= 24) and t < 60 and t % 10:
            continue
        say("t=%d" % t)
        for tag, _, _ in fams:
            r = rows[tag][t]
            say("   %-12s cost dict %.6f weights %.6f resid-form %.6f | spg f %.8f nfeval %d flags %d res %.3e | qp mean %.1f max %d parked %d | "
                "nnz C %d Z %d min row-max C %.3f moved %s"
                % (tag, r["c_d"], r["c_w"], r["rc"], r["f"], r["nfe"], r["flags"], r["res"], r["qp_mean"], r["qp_max"],
                   r["parked"], r["nnzC"], r["nnzZ"], r["cmax"], r["moved"]))
            sc = r["sc"]
            say("        lambda %.6f delta %.6e dd %.4e ainv %.6e alpha(BB after) %.4e s1d %.8e a1 %.8e a2 %.8e  (2 n delta = %.8e vs a1 - 2 s1d = %.8e)"
                % (sc["lambda"], sc["delta"], sc["dd"], sc["ainv"], sc["alpha"], sc["s1d"], sc["a1"], sc["a2"],
                   2 * N * sc["delta"], sc["a1"] - 2 * sc["s1d"]))


if __name__ == "__main__":
    main()
