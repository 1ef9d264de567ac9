#!/usr/bin/env python
"""How often does the 250-iteration run on bench.py's `parity_converged` problem end at the
oracle's optimum when the START is perturbed at float32-rounding size?

For each of several seeds the start dictionary is multiplied by (1 + eps * xi) and renormalised;
the HIP path then runs 250 outer iterations in float64, in float32 with the column-split
row-local kernel and in float32 without it.  The end points (residual-form reconstruction
error, computed on the host in float64) are compared with the oracle's unperturbed end point
from tests/golden/converged_1500.npz.  A kernel that is wrong shows as a family that misses
the oracle's optimum far more often than the others; a trajectory that is sensitive at
float32 size shows as the same miss rate in every family, float64 included.

Writes gpurun_out/diverge_ensemble.log."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "matrix-factorization-case-studies_amd"))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from convex_dim_red import _backend  # noqa: E402

N, K, T = 1500, 32, 250


def main():
    warnings.simplefilter("ignore")
    out = open(os.path.join(ROOT, "gpurun_out", "diverge_ensemble.log"), "w")

    def say(s):
        print(s, flush=True)
        out.write(s + "\n")
        out.flush()

    fx = np.load(os.path.join(ROOT, "tests", "golden", "converged_1500.npz"))
    o_end, o_arg = float(fx["oracle_reconstruction_error"]), fx["oracle_argmax"]
    X = bench.synthetic_rows(0, N).astype(np.float64)
    X32 = X.astype(np.float32)
    C0, Z0 = bench.start_factors(N, K)
    dkw = dict(max_iterations=1)
    families = [("float64", {}), ("float32", dict(row_local_acc64=1)), ("float32", dict(row_local_acc64=0)),
                ("float32", dict(row_local_acc64=0, row_local_split=0))]
    seeds = list(range(int(os.environ.get("ENS_SEEDS", "12"))))
    for eps in (1e-7, 1e-12):
        for dtype, opts in families:
            for name, v in opts.items():
                _backend.set_option(name, v)
            miss = 0
            rels = []
            with _backend.Context(dtype=dtype) as ctx:
                ctx.set_data(X32 if dtype == "float32" else X)
                for s in seeds:
                    rs = np.random.RandomState(100 + s)
                    C = C0 * (1 + eps * rs.standard_normal(C0.shape))
                    C /= C.sum(axis=1, keepdims=True)
                    ctx.set_state(C, Z0, np.ones(K))
                    ctx.prepare()
                    ctx.outer_iterations(T, dkw, {})
                    hC, hZ, _ = ctx.get_state()
                    rec = 0.5 * np.linalg.norm(X - hZ.dot(hC.dot(X))) ** 2 / N
                    rel = abs(rec - o_end) / o_end
                    rels.append(rel)
                    same = np.array_equal(hC.argmax(axis=1), o_arg)
                    miss += (rel > 1e-4) or not same
                    if rel > 1e-4 and eps == 1e-7:
                        # a plateau or another optimum?  750 more iterations
                        ctx.outer_iterations(750, dkw, {})
                        hC, hZ, _ = ctx.get_state()
                        rec2 = 0.5 * np.linalg.norm(X - hZ.dot(hC.dot(X))) ** 2 / N
                        say("    seed %d: %.6f after 250 iterations, %.6f after 1000 (oracle 250: %.6f)" % (s, rec, rec2, o_end))
            _backend.set_option("row_local_split", 1)
            _backend.set_option("row_local_acc64", 1)
            say("eps %.0e %-8s %-22s missed the oracle's optimum in %2d of %d starts; rel diffs: %s"
                % (eps, dtype, opts, miss, len(seeds), " ".join("%.1e" % r for r in rels)))


if __name__ == "__main__":
    main()
