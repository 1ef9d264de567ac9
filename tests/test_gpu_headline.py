"""The kernels bench.py TIMES, iterated against the oracle (needs an MI355X).

The product picks its pass kernels by shard size: from 32 768 rows per GPU on it runs
`k_row_local_f32_dma` + `k_reduce_rows_f32<1, 4>` (float32 data) and the float64 wave-streaming /
matrix-core kernels the headline benchmark runs; every other oracle comparison in tests/ is
smaller and runs the short-shard kernels (`k_row_local_f32_blk`).  Round 2's long-run divergence
lived exactly in the difference between two row-local kernels, so the timed mix gets its own
multi-iteration oracle test here: the first 40 000 rows of the headline workload, k = 32, bench.py's
start, three production outer iterations (reference archetypal_analysis.py:534-670), against
tests/golden/headline_40000.npz -- the oracle's run of the same problem with the cost after every
update, arg-max, supports, the largest dictionary entries, a subset of weight rows and the oracle's
OWN response to a one-ulp and to a float32-sized perturbation of X (written by
oracle/gen_headline_fixture.py; the oracle needs about 20 s of CPU for it).

Yardsticks (computed, stored in the fixture): float64 legs are held to 20 x the one-ulp twin,
float32 legs to 20 x the larger of two float32-sized twins of the oracle -- the data moved by 6e-8
relative, and the non-data operands of the four big contractions (C, Z, C X, X'Z) rounded to
float32, which is what the float32 mode feeds the matrix cores (oracle.operand_rounding; C X is an
average over thousands of samples, so rounding IT moves it far more than noise on the samples:
weights of the first iteration 1.2e-5 against 1.3e-7) -- each with a floor that says what it is:
  cost              float64: 1e-10 (summation order of 1.6e8 products); float32: the noise of the
                    trace form, 8 eps32 tr(XX')/n / cost (the cost cancels
                    tr(XX')/n down to the residual)
  weights           1e-6, the per-sample QP's own stopping tolerance (spg.py:388-390)
  dictionary        float64 1e-12, float32 5e-7 (one projected step of length alpha ~ 1 along a
                    gradient with fp32 relative error ~1e-7)
arg-max exact in both dtypes; support sizes exact in float64."""
import warnings

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

N, K, T = 40000, 32, 3
SPG_KW, QP_KW = dict(max_iterations=1), {}


@pytest.fixture(scope="module")
def problem():
    import bench
    from convex_dim_red import _backend
    _backend.require_gpu()
    X = bench.synthetic_rows(0, N)                      # float32 values, as the benchmark holds them
    C0, Z0 = bench.start_factors(N, K)
    return X, C0, Z0, load_golden("headline_40000")


def _limits(fx, dtype, X):
    # float64: the one-ulp twin; float32: the larger of the two float32-sized twins (data moved by
    # 6e-8 relative; operands of the four contractions rounded to float32)
    class _Twin(object):
        def __getitem__(self, key):
            if dtype == "float64":
                return fx["twin" + key]
            return np.maximum(fx["f32" + key], fx["op32" + key])
    tw = _Twin()
    noise = 0.0
    if dtype == "float32":
        tr_n = float((X.astype(np.float64) ** 2).sum()) / N
        noise = 8 * 6e-8 * tr_n
    return dict(
        cost=lambda t, want: max((1e-10 if dtype == "float64" else 0.0) * want, noise,
                                 20 * tw["_cost_rel"][t] * want),
        cost_d=lambda t, want: max((1e-10 if dtype == "float64" else 0.0) * want, noise,
                                   20 * tw["_cost_dictionary_rel"][t] * want),
        Z=lambda t: max(1e-6, 20 * tw["_Z_rows_maxdiff"][t]),
        C=lambda t: max(1e-12 if dtype == "float64" else 5e-7, 20 * tw["_C_top_maxdiff"][t]),
        Zsum=lambda t: max(1e-6 * np.sqrt(N), 20 * tw["_Z_colsum_maxdiff"][t]),
        Csq=lambda t: max(1e-12 if dtype == "float64" else 5e-7, 20 * tw["_C_rowsq_maxrel"][t]),
    )


def _compare(tag, t, fx, lim, dtype, cost_d, cost, C, Z):
    rows = fx["rows"]
    top = fx["C_top_idx"][t]
    got_top = np.take_along_axis(C, top, axis=1)
    d = dict(cost=abs(cost - fx["cost"][t]), cost_d=abs(cost_d - fx["cost_dictionary"][t]),
             Z=np.abs(Z[rows] - fx["Z_rows"][t]).max(), C=np.abs(got_top - fx["C_top_val"][t]).max(),
             Zsum=np.abs(Z.sum(axis=0) - fx["Z_colsum"][t]).max(),
             Csq=(np.abs((C * C).sum(axis=1) - fx["C_rowsq"][t]) / fx["C_rowsq"][t]).max())
    bound = dict(cost=lim["cost"](t, fx["cost"][t]), cost_d=lim["cost_d"](t, fx["cost_dictionary"][t]),
                 Z=lim["Z"](t), C=lim["C"](t), Zsum=lim["Zsum"](t), Csq=lim["Csq"](t))
    print("headline %s %s iteration %d: " % (tag, dtype, t + 1)
          + ", ".join("%s %.2e (bound %.2e)" % (k, d[k], bound[k]) for k in sorted(d)))
    for k in d:
        assert d[k] <= bound[k], (tag, dtype, t, k, d[k], bound[k])
    assert np.array_equal(C.argmax(axis=1), fx["argmax"][t]), (tag, dtype, t)
    if dtype == "float64":
        assert np.array_equal((C > 1e-15).sum(axis=1), fx["C_support"][t]), (tag, dtype, t)
    else:
        # float32: an entry whose projected value is a rounding-sized quantity may fall on either
        # side of the threshold; the sizes of the supports (~hundreds) agree to a few entries
        assert np.abs((C > 1e-15).sum(axis=1) - fx["C_support"][t]).max() <= 2 + 0.01 * fx["C_support"][t].max()
    assert np.all(C >= 0) and np.all(Z >= 0)
    assert np.allclose(C.sum(axis=1), 1, rtol=0, atol=1e-12)
    assert np.allclose(Z.sum(axis=1), 1, rtol=0, atol=1e-12)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_timed_kernel_mix_against_the_oracle(problem, dtype):
    from convex_dim_red import _backend
    X, C0, Z0, fx = problem
    Xh = X if dtype == "float32" else X.astype(np.float64)
    lim = _limits(fx, dtype, X)
    with _backend.Context(dtype=dtype) as ctx:
        ctx.set_data(Xh)
        ctx.set_state(C0, Z0, np.ones(K))
        ctx.prepare()
        # the passes this size selects are the ones the benchmark times
        ctx.gemm_timing(True)
        for t in range(T):
            costs = ctx.outer_iterations(1, SPG_KW, QP_KW)
            C, Z, _ = ctx.get_state()
            _compare("outer_iterations", t, fx, lim, dtype, costs[0], costs[1], C, Z)
        _, n_reduce, _, n_local = ctx.gemm_timing(False)
        assert n_reduce >= 2 * T and n_local >= 2 * T
        want = {"float32": ("k_reduce_rows_f32<1,4>", "k_row_local_f32_dma<8>"),
                "float64": ("k_reduce_rows_f64_mfma<2>", "k_row_local_f64_ws<2>")}[dtype]
        assert ctx.pass_kernels() == want                 # what bench.py's 100 000 rows select too
        # the loop the estimators run (aa_iterate) from the same start: same end state
        ctx.set_state(C0, Z0, np.ones(K))
        cost0 = ctx.prepare()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            costs, st = ctx.iterate(cost0, T, 0.0, "abs_delta_f", False, True, True, SPG_KW, QP_KW)
        assert st.n_iter == T - 1
        C2, Z2, _ = ctx.get_state()
        _compare("aa_iterate", T - 1, fx, lim, dtype, costs[-2], costs[-1], C2, Z2)
        # two drivers, one set of kernels (the sample order of the QP, taken from the previous
        # update's pass counts, decides only which wave takes a sample)
        print("aa_iterate against outer_iterations: bit-identical %s, max |dC| %.1e, max |dZ| %.1e"
              % (np.array_equal(C2, C) and np.array_equal(Z2, Z), np.abs(C2 - C).max(), np.abs(Z2 - Z).max()))
        assert np.abs(C2 - C).max() <= 1e-12 and np.abs(Z2 - Z).max() <= 1e-9
