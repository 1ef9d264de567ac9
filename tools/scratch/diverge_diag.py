#!/usr/bin/env python
"""Where does the HIP path leave the oracle's trajectory on bench.py's `parity_converged`
problem (first 1500 rows of the headline workload, k = 32, 250 outer iterations)?

Three experiments, oracle side computed once:
  A. whole trajectories under different library options, both dtypes: end-point distance and
     the first outer iteration at which the cost leaves the oracle's curve;
  B. the stateless pieces on the ORACLE's state of every iteration: the per-sample QPs
     (qp_batch on the oracle's A, B, Z0: iterates and pass counts) and one dictionary update
     (fresh context) -- a kernel that is wrong on some state shows up at that iteration;
  C. one outer iteration of a long-lived context started from the oracle's state.

Writes gpurun_out/diverge_diag.log.  Test infrastructure (imports the oracle)."""
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "matrix-factorization-case-studies_amd"))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from convex_dim_red import _backend  # noqa: E402
from convex_dim_red import archetypal_analysis as aa  # noqa: E402
from oracle import aa_oracle as orc  # noqa: E402

N, K, T = int(os.environ.get("DIAG_N", "1500")), 32, int(os.environ.get("DIAG_T", "250"))
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
LOG = open(os.path.join(OUT, os.environ.get("DIAG_LOG", "diverge_diag.log")), "w")


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
    C0, Z0 = bench.start_factors(N, K)
    alpha = np.ones(K)
    trX = float((X * X).sum())
    dkw = dict(max_iterations=1)

    # ---------------- oracle, stepwise (bit-identical to one long iterate_aa run)
    t0 = time.perf_counter()
    states = [(C0, Z0)]
    qp_in = []
    o_iters = []
    o_costs = []
    C, Z = C0, Z0
    for t in range(T):
        ZtZ = Z.T.dot(Z)
        XtZ = X.T.dot(Z)
        XXtZ = X.dot(XtZ)
        C1 = orc.update_aa_dictionary(X, C, alpha, trX, XXtZ, ZtZ, **dkw)[0]
        CX = C1.dot(X)
        CXXt = CX.dot(X.T)
        CXXtCt = CX.dot(CX.T)
        Z1, it = orc.update_kernel_aa_weights(Z, alpha, CXXt, CXXtCt, return_iters=True)
        qp_in.append((CXXtCt, CXXt, Z))
        o_iters.append(it)
        C, Z = C1, Z1
        states.append((C, Z))
        o_costs.append(rec_err(X, Z, C))
    say("oracle: %d iterations in %.1f s, final residual-form error %.10f" % (T, time.perf_counter() - t0, o_costs[-1]))
    # cross-check against the long run the bench uses
    oZ, oC, _, ocost, _, _, odeltas = orc.iterate_aa(X, Z0, C0, alpha, trace_XXt=trX, tolerance=0, max_iterations=T,
                                                     dictionary_solver_kwargs=dkw, require_monotonic_cost_decrease=False)
    say("oracle long run == stepwise: C %s Z %s" % (np.array_equal(oC, C), np.array_equal(oZ, Z)))
    o_end = rec_err(X, oZ, oC)
    o_arg = oC.argmax(axis=1)
    # twin: 1-ulp perturbation of X
    rs = np.random.RandomState(7)
    Xp = X * (1 + 2e-16 * rs.standard_normal(X.shape))
    tZ, tC = orc.iterate_aa(Xp, Z0, C0, alpha, trace_XXt=float((Xp * Xp).sum()), tolerance=0, max_iterations=T,
                            dictionary_solver_kwargs=dkw, require_monotonic_cost_decrease=False)[:2]
    twin = abs(rec_err(X, tZ, tC) - o_end) / o_end
    say("oracle twin (X 1-ulp perturbed): rel diff of end point %.3e, argmax equal %s"
        % (twin, np.array_equal(tC.argmax(axis=1), o_arg)))

    # ---------------- A. whole trajectories under options
    def run(dtype, opts):
        for name, v in opts.items():
            _backend.set_option(name, v)
        try:
            with _backend.Context(dtype=dtype) as ctx:
                ctx.set_data(X.astype(np.float32) if dtype == "float32" else X)
                ctx.set_state(C0, Z0, alpha)
                ctx.prepare()
                costs = ctx.outer_iterations(T, dkw, {})
                hC, hZ, _ = ctx.get_state()
        finally:
            for name in opts:
                _backend.set_option(name, DEFAULTS[name])
        return hC, hZ, costs

    DEFAULTS = dict(qp_mode=0, qp_quad_cap=0, qp_pass_cap=24, row_local_split=1, row_local_variant=-1, f64_mfma=1,
                    proj_small=1, proj_mode=0, qp_sort=1, fuse_finalize=1, pq_blocks=128, qp_quad_occ=3)
    configs = [{}, dict(qp_mode=1), dict(qp_mode=2), dict(qp_mode=3), dict(qp_mode=4),
               dict(qp_quad_cap=1), dict(qp_quad_cap=8), dict(qp_quad_cap=1000),
               dict(qp_mode=2, qp_pass_cap=1000), dict(row_local_split=0), dict(f64_mfma=0), dict(f64_mfma=3),
               dict(proj_small=0), dict(proj_mode=1), dict(qp_sort=0), dict(fuse_finalize=0), dict(pq_blocks=16),
               dict(qp_quad_occ=2)]
    # oracle cost curve in trace form for the first-departure search: use residual-form checkpoints instead
    o_curve = np.array(o_costs)
    for dtype in ("float64", "float32"):
        for opts in configs:
            try:
                hC, hZ, costs = run(dtype, opts)
            except Exception as e:                      # noqa: BLE001
                say("A %-8s %-32s ERROR %s" % (dtype, opts, e))
                continue
            end = rec_err(X, hZ, hC)
            h_curve = costs[1::2]
            # trace-form HIP cost vs residual-form oracle cost: equal up to rounding in float64
            rel = np.abs(h_curve - o_curve) / o_curve
            first = {thr: (int(np.argmax(rel > thr)) if np.any(rel > thr) else -1) for thr in (1e-6, 1e-4, 1e-3)}
            say("A %-8s %-32s end rel %.3e argmax_eq %-5s first>1e-6 @%d  >1e-4 @%d  >1e-3 @%d  max rel %.2e"
                % (dtype, json.dumps(opts), abs(end - o_end) / o_end,
                   np.array_equal(hC.argmax(axis=1), o_arg), first[1e-6], first[1e-4], first[1e-3], rel.max()))

    # ---------------- B. stateless pieces on the oracle's states
    say("B: QP (default mode) and dictionary update on the oracle's state of every iteration")
    worst_qp, worst_d = 0.0, 0.0
    for t in range(T):
        A, B, Zin = qp_in[t]
        Zh, ith = _backend.qp_batch(A, B, Zin, "kn", return_iters=True)
        dz = np.abs(Zh - states[t + 1][1]).max()
        nbad = int((ith != o_iters[t]).sum())
        C_t, Z_t = states[t]
        ZtZ = Z_t.T.dot(Z_t)
        XXtZ = X.dot(X.T.dot(Z_t))
        Ch = aa._update_aa_dictionary(X, C_t, alpha, trX, XXtZ, ZtZ, **dkw)
        dc = np.abs(Ch - states[t + 1][0]).max()
        worst_qp, worst_d = max(worst_qp, dz), max(worst_d, dc)
        if dz > 1e-7 or dc > 1e-9 or nbad or t % 25 == 0:
            rows = np.nonzero(np.abs(Zh - states[t + 1][1]).max(axis=1) > 1e-7)[0]
            say("B t=%3d  QP max|dZ| %.2e  pass counts differing %4d (oracle max %d mean %.1f)  rows off %s (oracle passes %s hip %s) | dict max|dC| %.2e"
                % (t, dz, nbad, o_iters[t].max(), o_iters[t].mean(), rows[:6].tolist(),
                   o_iters[t][rows[:6]].tolist(), ith[rows[:6]].tolist(), dc))
    say("B worst: QP %.3e, dictionary %.3e" % (worst_qp, worst_d))

    # ---------------- C. one outer iteration of a long-lived context from the oracle's state
    for dtype in ("float64", "float32"):
        worst = (0.0, 0.0)
        with _backend.Context(dtype=dtype) as ctx:
            ctx.set_data(X.astype(np.float32) if dtype == "float32" else X)
            for t in range(T):
                ctx.set_state(states[t][0], states[t][1], alpha)
                ctx.prepare()
                ctx.outer_iterations(1, dkw, {})
                hC, hZ, _ = ctx.get_state()
                dc = np.abs(hC - states[t + 1][0]).max()
                dz = np.abs(hZ - states[t + 1][1]).max()
                worst = (max(worst[0], dc), max(worst[1], dz))
                if (dtype == "float64" and (dc > 1e-8 or dz > 1e-6)) or t % 50 == 0:
                    say("C %s t=%3d max|dC| %.2e max|dZ| %.2e" % (dtype, t, dc, dz))
        say("C %s worst dC %.3e dZ %.3e" % (dtype, worst[0], worst[1]))


if __name__ == "__main__":
    main()
