#!/usr/bin/env python
"""Writes tests/golden/converged_1500.npz: the oracle's long run on bench.py's
`parity_converged` problem, so that the GPU box can check the HIP path against it without a
40 s CPU run.  TEST INFRASTRUCTURE (see aa_oracle.py's header).

Problem: the first 1500 rows of the headline workload (bench.synthetic_rows, float32 values),
k = 32, init='random' start from RandomState(1) (bench.start_factors), delta = 0,
dictionary_solver_kwargs = {max_iterations: 1}, default weights solver, 250 outer iterations with
tolerance 0 -- reference archetypal_analysis.py:534-670 through oracle.iterate_aa.

Stored (data only):
  oracle_reconstruction_error   0.5 ||X - Z C X||_F^2 / n at the end point (residual form, float64)
  oracle_cost                   the trace-form cost the loop itself reports at the end point
  oracle_argmax                 argmax of every dictionary row at the end point
  cost_curve                    trace-form cost after every outer iteration (checkpoints)
  twin_rel                      |end point(X) - end point(X (1 + 2e-16 xi))| / end point: the oracle's
                                own sensitivity to a perturbation of one unit in the last place
  ens_eps, ens_rel, ens_argmax_equal
                                end points of the oracle under float32-sized perturbations of the
                                data, X (1 + eps xi): how far an exact-arithmetic run moves when
                                its input is disturbed at the size of float32 rounding
Usage: python oracle/gen_converged_fixture.py [n_ensemble_per_eps]   (about 1 min per oracle run)"""
import os
import sys
import warnings
from concurrent.futures import ProcessPoolExecutor

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")          # before NumPy: runs go side by side
os.environ.setdefault("OMP_NUM_THREADS", "1")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N, K, T = 1500, 32, 250


def run(job):
    seed, eps = job
    import bench
    from oracle import aa_oracle as orc
    warnings.simplefilter("ignore")
    X = bench.synthetic_rows(0, N).astype(np.float64)
    C0, Z0 = bench.start_factors(N, K)
    Xp = X
    if eps > 0:
        Xp = X * (1 + eps * np.random.RandomState(seed).standard_normal(X.shape))
    Z, C, _, cost, _, _, deltas = orc.iterate_aa(
        Xp, Z0, C0, np.ones(K), trace_XXt=float((Xp * Xp).sum()), tolerance=0, max_iterations=T,
        dictionary_solver_kwargs=dict(max_iterations=1), require_monotonic_cost_decrease=False)
    rec = 0.5 * np.linalg.norm(X - Z.dot(C.dot(X))) ** 2 / N
    return dict(seed=seed, eps=eps, rec=rec, cost=cost, argmax=C.argmax(axis=1),
                curve=cost - np.cumsum(np.asarray(deltas)[::-1])[::-1] + np.asarray(deltas))


def main():
    n_ens = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    jobs = [(0, 0.0), (7, 2e-16)]
    for eps in (6e-8, 1e-6):
        jobs += [(s, eps) for s in range(1, n_ens + 1)]
    with ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        res = list(ex.map(run, jobs))
    base, twin, ens = res[0], res[1], res[2:]
    out = os.path.join(ROOT, "tests", "golden", "converged_1500.npz")
    np.savez_compressed(
        out,
        what=np.array("oracle.iterate_aa on bench.synthetic_rows(0,1500), k=32, 250 outer iterations"),
        oracle_reconstruction_error=base["rec"], oracle_cost=base["cost"],
        oracle_argmax=base["argmax"].astype(np.int32), cost_curve=base["curve"],
        twin_rel=abs(twin["rec"] - base["rec"]) / base["rec"],
        twin_argmax_equal=np.array_equal(twin["argmax"], base["argmax"]),
        ens_eps=np.array([r["eps"] for r in ens]),
        ens_rel=np.array([abs(r["rec"] - base["rec"]) / base["rec"] for r in ens]),
        ens_argmax_equal=np.array([np.array_equal(r["argmax"], base["argmax"]) for r in ens]))
    print("wrote", out)
    print("end point %.12f, twin rel %.3e" % (base["rec"], abs(twin["rec"] - base["rec"]) / base["rec"]))
    for r in ens:
        print("eps %.0e seed %d: rel %.3e argmax equal %s" % (r["eps"], r["seed"], abs(r["rec"] - base["rec"]) / base["rec"],
                                                           np.array_equal(r["argmax"], base["argmax"])))


if __name__ == "__main__":
    main()
