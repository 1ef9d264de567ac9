#!/usr/bin/env python
"""Writes tests/golden/headline_40000.npz: the oracle's first outer iterations on the first
40 000 rows of the headline workload, so that the GPU box can put the kernels bench.py TIMES
under the oracle.  TEST INFRASTRUCTURE (see aa_oracle.py's header).

Why 40 000 rows: the product picks its pass kernels by shard size, and from 32 768 rows per GPU
on it runs the ones the benchmark runs (k_row_local_f32_dma, k_reduce_rows_f32<1,4>, the float64
wave-streaming kernels).  Every other oracle comparison in tests/ is smaller and therefore runs
the short-shard kernels.

Problem: bench.synthetic_rows(0, 40000) (float32 values), k = 32, init='random' start from
RandomState(1) (bench.start_factors), delta = 0, dictionary_solver_kwargs = {max_iterations: 1},
default weights solver, T = 3 outer iterations with tolerance 0 -- reference
archetypal_analysis.py:534-670 through oracle.iterate_aa, run one outer iteration at a time so
that the state after every iteration can be stored.

Stored (data only, float64):
  cost                 trace-form cost after each outer iteration              [T]
  cost_dictionary      ... and after the dictionary update inside it            [T]
  argmax               argmax of every dictionary row after each iteration      [T][k]
  rows                 the sample rows whose weights are stored (every 61st)
  Z_rows               weights of those samples after each iteration            [T][len(rows)][k]
  C_support            number of non-zeros of every dictionary row              [T][k]
  C_top_idx, C_top_val the 8 largest entries of every dictionary row            [T][k][8]
  Z_colsum, C_rowsq    column sums of Z and sum of squares of C's rows          [T][k]
  twin_*               the same run on X (1 + 2e-16 xi): the oracle's own response to a one-ulp
                       perturbation of the data, the yardstick for everything above
  f32_*                ... and on X (1 + 6e-8 xi), the float32-sized perturbation of the data
  op32_*               ... and with the non-data operand of every big contraction against X rounded
                       to float32 (oracle.operand_rounding): C, Z, C X and X'Z as the float32 mode of
                       the HIP path feeds them to the matrix cores.  C X is an average over thousands of
                       samples, so rounding IT to float32 moves it far more than float32-sized noise on
                       the samples does -- the weights of the first iteration, where the archetypes
                       of a random start are nearly equal and the per-sample QPs ill conditioned,
                       respond 700 x more strongly (1e-4 against 1.3e-7)
Usage: python oracle/gen_headline_fixture.py      (a few minutes of CPU, ~6 GB)"""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N, K, T, STRIDE, TOP = 40000, 32, 3, 61, 8


def run(X, eps, seed, operand_dtype=None):
    import bench
    from oracle import aa_oracle as orc
    if operand_dtype is not None:
        with orc.operand_rounding(operand_dtype):
            return run(X, eps, seed)
    C, Z = bench.start_factors(N, K)
    Xp = X
    if eps > 0:
        Xp = X * (1 + eps * np.random.RandomState(seed).standard_normal(X.shape))
    trace = float((Xp * Xp).sum())
    rows = np.arange(0, N, STRIDE)
    out = dict(cost=[], cost_dictionary=[], argmax=[], Z_rows=[], C_support=[], C_top_idx=[],
               C_top_val=[], Z_colsum=[], C_rowsq=[])
    for t in range(T):
        log = []
        Z, C, _, cost, _, _, deltas = orc.iterate_aa(
            Xp, Z, C, np.ones(K), trace_XXt=trace, tolerance=0, max_iterations=1,
            dictionary_solver_kwargs=dict(max_iterations=1), require_monotonic_cost_decrease=False,
            cost_log=log)
        out["cost_dictionary"].append(dict(log)["dictionary"])
        out["cost"].append(cost)
        out["argmax"].append(C.argmax(axis=1))
        out["Z_rows"].append(Z[rows].copy())
        out["C_support"].append((C > 1e-15).sum(axis=1))
        top = np.argsort(-C, axis=1, kind="stable")[:, :TOP]
        out["C_top_idx"].append(top)
        out["C_top_val"].append(np.take_along_axis(C, top, axis=1))
        out["Z_colsum"].append(Z.sum(axis=0))
        out["C_rowsq"].append((C * C).sum(axis=1))
    return {k: np.asarray(v) for k, v in out.items()}, rows


def main():
    import bench
    warnings.simplefilter("ignore")
    t0 = time.time()
    X = bench.synthetic_rows(0, N).astype(np.float64)
    print("data %.0f s" % (time.time() - t0), flush=True)
    base, rows = run(X, 0.0, 0)
    print("oracle run %.0f s, costs %s" % (time.time() - t0, base["cost"]), flush=True)
    twin, _ = run(X, 2e-16, 7)
    f32, _ = run(X, 6e-8, 8)
    op32, _ = run(X, 0.0, 0, np.float32)
    print("twins %.0f s" % (time.time() - t0), flush=True)
    out = os.path.join(ROOT, "tests", "golden", "headline_40000.npz")
    keep = dict(base)
    for tag, other in (("twin", twin), ("f32", f32), ("op32", op32)):
        keep[tag + "_cost_rel"] = np.abs(other["cost"] - base["cost"]) / base["cost"]
        keep[tag + "_cost_dictionary_rel"] = (np.abs(other["cost_dictionary"] - base["cost_dictionary"])
                                              / base["cost_dictionary"])
        keep[tag + "_Z_rows_maxdiff"] = np.abs(other["Z_rows"] - base["Z_rows"]).reshape(T, -1).max(axis=1)
        keep[tag + "_C_top_maxdiff"] = np.abs(other["C_top_val"] - base["C_top_val"]).reshape(T, -1).max(axis=1)
        keep[tag + "_Z_colsum_maxdiff"] = np.abs(other["Z_colsum"] - base["Z_colsum"]).max(axis=1)
        keep[tag + "_C_rowsq_maxrel"] = (np.abs(other["C_rowsq"] - base["C_rowsq"]) / base["C_rowsq"]).max(axis=1)
        keep[tag + "_argmax_equal"] = np.array([np.array_equal(a, b) for a, b in zip(other["argmax"], base["argmax"])])
        keep[tag + "_support_equal"] = np.array([np.array_equal(a, b) for a, b in zip(other["C_support"], base["C_support"])])
    np.savez_compressed(
        out, what=np.array("oracle.iterate_aa on bench.synthetic_rows(0,40000), k=32, 3 outer iterations"),
        rows=rows.astype(np.int32), **keep)
    for k in sorted(keep):
        if k.startswith(("twin_", "f32_", "op32_")):
            print(k, keep[k])
    print("wrote %s (%d bytes)" % (out, os.path.getsize(out)))


if __name__ == "__main__":
    main()
