#!/usr/bin/env python
"""Does the float32 run stay with the oracle when the fp32 accumulation chains of the row-local
pass are shorter?  bench.py's parity_converged problem, float32, column chunk of the block-tiled
row-local kernel forced to 4096 (no split) / 512 / 128 / 32; per run: the step length the
dictionary line search takes at outer iteration 15 (oracle / float64: 0.3423), the number of
non-zeros of the dictionary at iteration 20 (oracle: 197) and the end point after 250.
Writes gpurun_out/diverge_chunk.log."""
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
    log = open(os.path.join(ROOT, "gpurun_out", "diverge_chunk.log"), "w")

    def say(s):
        print(s, flush=True)
        log.write(s + "\n")
        log.flush()

    fx = np.load(os.path.join(ROOT, "tests", "golden", "converged_1500.npz"))
    o_end, o_arg = float(fx["oracle_reconstruction_error"]), fx["oracle_argmax"]
    X = bench.synthetic_rows(0, N).astype(np.float64)
    X32 = X.astype(np.float32)
    C0, Z0 = bench.start_factors(N, K)
    dkw = dict(max_iterations=1)
    for dtype, variant, chunk, a64 in (("float64", -1, 0, 1), ("float32", -1, 0, 1), ("float32", -1, 0, 0),
                                       ("float32", 4, 4096, 0), ("float32", 4, 128, 0), ("float32", 7, 32, 0)):
        _backend.set_option("row_local_variant", variant)
        _backend.set_option("row_local_chunk", chunk)
        _backend.set_option("row_local_acc64", a64)
        with _backend.Context(dtype=dtype) as ctx:
            ctx.set_data(X32 if dtype == "float32" else X)
            for seed in range(6):
                C = C0
                if seed:
                    C = C0 * (1 + 1e-7 * np.random.RandomState(100 + seed).standard_normal(C0.shape))
                    C /= C.sum(axis=1, keepdims=True)
                ctx.set_state(C, Z0, np.ones(K))
                ctx.prepare()
                lam15 = None
                for t in range(21):
                    ctx.dictionary_update(**dkw)
                    if t == 15:
                        lam15 = ctx.spg_scalars()["lambda"]
                    ctx.weights_update()
                nnz20 = int((ctx.get_state()[0] > 0).sum())
                ctx.outer_iterations(T - 21, dkw, {})
                hC, hZ, _ = ctx.get_state()
                rec = 0.5 * np.linalg.norm(X - hZ.dot(hC.dot(X))) ** 2 / N
                say("%-8s acc64 %d variant %2d chunk %4d seed %d: lambda(t=15) %.4f  nnz C(t=20) %3d  end rel %.2e argmax equal %s"
                    % (dtype, a64, variant, chunk, seed, lam15, nnz20, abs(rec - o_end) / o_end,
                       np.array_equal(hC.argmax(axis=1), o_arg)))
    _backend.set_option("row_local_variant", -1)
    _backend.set_option("row_local_chunk", 0)
    _backend.set_option("row_local_acc64", 1)


if __name__ == "__main__":
    main()
