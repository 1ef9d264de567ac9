#!/usr/bin/env python
"""Second round of the long-run divergence search: float32 mode, column-split row-local kernel.

1. the two pass kernels on their own against float64 NumPy (aa_pass_row_local /
   aa_pass_reduce_rows), split on/off, operands taken from the run itself;
2. whole cost curves (float32 split on / off, qp_mode 1, float64) saved for offline comparison;
3. one outer iteration from the oracle's state at every iteration, float32, split on / off.
Writes gpurun_out/diverge_diag2.log and gpurun_out/diverge_curves.npz."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "matrix-factorization-case-studies_amd"))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from convex_dim_red import _backend  # noqa: E402
from oracle import aa_oracle as orc  # noqa: E402

N, K, T = 1500, 32, 250
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
LOG = open(os.path.join(OUT, "diverge_diag2.log"), "w")


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    LOG.write(s + "\n")
    LOG.flush()


def rec_err(X, Z, C):
    return 0.5 * np.linalg.norm(X - Z.dot(C.dot(X))) ** 2 / X.shape[0]


def main():
    warnings.simplefilter("ignore")
    X = bench.synthetic_rows(0, N).astype(np.float64)
    X32 = X.astype(np.float32)
    C0, Z0 = bench.start_factors(N, K)
    alpha = np.ones(K)
    trX = float((X * X).sum())
    dkw = dict(max_iterations=1)

    # ---- 1. pass kernels on their own
    rs = np.random.RandomState(3)
    operands = {"C0 X": C0.dot(X), "random": rs.standard_normal((K, X.shape[1])),
                "one-hot rows of X": X[rs.choice(N, K, replace=False)]}
    for split in (1, 0):
        _backend.set_option("row_local_split", split)
        for dtype in ("float32", "float64"):
            with _backend.Context(dtype=dtype) as ctx:
                ctx.set_data(X32 if dtype == "float32" else X)
                for name, B in operands.items():
                    got = ctx.pass_row_local(B)
                    want = X.dot(B.T)
                    scale = np.abs(X).dot(np.abs(B).T)              # sum |x||b|: the rounding yardstick
                    err = np.abs(got - want)
                    say("1 row_local split=%d %-8s B=%-18s max|err| %.3e  max err/sum|x||b| %.3e  rms rel-to-yardstick %.3e  mean signed %.3e"
                        % (split, dtype, name, err.max(), (err / scale).max(), np.sqrt(((err / scale) ** 2).mean()),
                           ((got - want) / scale).mean()))
                A = Z0 if True else None
                got = ctx.pass_reduce_rows(A)
                want = A.T.dot(X)
                scale = np.abs(A).T.dot(np.abs(X))
                err = np.abs(got - want)
                say("1 reduce_rows        %-8s A=Z0                  max|err| %.3e  max err/sum|a||x| %.3e  mean signed %.3e"
                    % (dtype, err.max(), (err / scale).max(), ((got - want) / scale).mean()))
    _backend.set_option("row_local_split", 1)

    # ---- oracle states
    states = [(C0, Z0)]
    C, Z = C0, Z0
    o_cost = []
    for t in range(T):
        ZtZ = Z.T.dot(Z)
        XXtZ = X.dot(X.T.dot(Z))
        C1 = orc.update_aa_dictionary(X, C, alpha, trX, XXtZ, ZtZ, **dkw)[0]
        CX = C1.dot(X)
        Z1 = orc.update_kernel_aa_weights(Z, alpha, CX.dot(X.T), CX.dot(CX.T))
        C, Z = C1, Z1
        states.append((C, Z))
        o_cost.append(rec_err(X, Z, C))
    o_cost = np.array(o_cost)
    say("oracle end %.10f" % o_cost[-1])

    # ---- 2. curves
    curves = {"oracle": o_cost}
    runs = [("f32_split1", "float32", dict(row_local_split=1)), ("f32_split0", "float32", dict(row_local_split=0)),
            ("f32_split1_qpmode1", "float32", dict(row_local_split=1, qp_mode=1)),
            ("f32_split0_qpmode1", "float32", dict(row_local_split=0, qp_mode=1)),
            ("f32_split1_qpmode2", "float32", dict(row_local_split=1, qp_mode=2)),
            ("f32_split0_qpmode2", "float32", dict(row_local_split=0, qp_mode=2)),
            ("f32_split1_rr256", "float32", dict(row_local_split=1, reduce_rows_blocks=256)),
            ("f32_split0_rr256", "float32", dict(row_local_split=0, reduce_rows_blocks=256)),
            ("f64", "float64", {})]
    defaults = dict(row_local_split=1, qp_mode=0, reduce_rows_blocks=512)
    for tag, dtype, opts in runs:
        for name, v in opts.items():
            _backend.set_option(name, v)
        with _backend.Context(dtype=dtype) as ctx:
            ctx.set_data(X32 if dtype == "float32" else X)
            ctx.set_state(C0, Z0, alpha)
            ctx.prepare()
            per_it = []
            for t in range(T):                           # one iteration at a time: residual-form cost of every iterate
                ctx.outer_iterations(1, dkw, {})
                per_it.append(ctx.reconstruction_cost())
            hC, hZ, _ = ctx.get_state()
        for name in opts:
            _backend.set_option(name, defaults[name])
        curves[tag] = np.array(per_it)
        rel = np.abs(curves[tag] - o_cost) / o_cost
        say("2 %-22s end rel %.3e (host residual form %.3e) argmax_eq %s; rel at t=5,10,15,20,30,50,100,150,200: %s"
            % (tag, rel[-1], abs(rec_err(X, hZ, hC) - o_cost[-1]) / o_cost[-1],
               np.array_equal(hC.argmax(axis=1), states[-1][0].argmax(axis=1)),
               " ".join("%.1e" % rel[i] for i in (5, 10, 15, 20, 30, 50, 100, 150, 200))))
    np.savez(os.path.join(OUT, "diverge_curves.npz"), **curves)

    # ---- 3. one step from the oracle's state, float32, split on/off
    for split in (1, 0):
        _backend.set_option("row_local_split", split)
        dcs, dzs = [], []
        with _backend.Context(dtype="float32") as ctx:
            ctx.set_data(X32)
            for t in range(T):
                ctx.set_state(states[t][0], states[t][1], alpha)
                ctx.prepare()
                ctx.outer_iterations(1, dkw, {})
                hC, hZ, _ = ctx.get_state()
                dcs.append(np.abs(hC - states[t + 1][0]).max())
                dzs.append(np.abs(hZ - states[t + 1][1]).max())
        dcs, dzs = np.array(dcs), np.array(dzs)
        say("3 float32 split=%d one-step errors: dC median %.2e max %.2e at t=%d; dZ median %.2e max %.2e at t=%d; dC t=10..20: %s"
            % (split, np.median(dcs), dcs.max(), dcs.argmax(), np.median(dzs), dzs.max(), dzs.argmax(),
               " ".join("%.1e" % v for v in dcs[10:21])))
    _backend.set_option("row_local_split", 1)


if __name__ == "__main__":
    main()
