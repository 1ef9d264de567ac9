"""Long-run parity on bench.py's `parity_converged` problem (needs an MI355X).

The first 1500 rows of the headline workload, k = 32, 250 outer iterations from bench.py's start
with the production settings, against the oracle's run of the same problem stored in
tests/golden/converged_1500.npz (written by oracle/gen_converged_fixture.py: end point, arg-max,
cost after every iteration, the oracle's own sensitivity to a one-ulp perturbation of the data
and to float32-sized perturbations).  Reference: archetypal_analysis.py:534-670.

History: round 2 ended with the float32 mode 1.0e-2 away from the oracle on this problem, with
another arg-max, and nothing under pytest noticed.  Cause: `x_r . p_i` is
a coherent sum, the fp32 rounding error of its running sum grows with the length of the
accumulation chain, and at outer iteration 15 the dictionary line search of this problem
interpolates its step from a cancelling difference (a1 - 2 s1d + a2 = -500.4 + 505.3) -- the
accumulated float32 error moved the step from 0.342 to 0.172, the next update accepted a full
step, the dictionary lost three quarters of its support and from that regime half the runs end
in another local optimum.  The float32 row-local kernel for shards below 32 768 rows now cuts
the fp32 chain every 32 columns and sums the pieces in float64.
"""
import os
import warnings

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

N, K, T = 1500, 32, 250


@pytest.fixture(scope="module")
def cdr():
    import convex_dim_red
    from convex_dim_red import _backend
    _backend.require_gpu()
    return convex_dim_red


@pytest.fixture(scope="module")
def problem():
    import bench
    X = bench.synthetic_rows(0, N).astype(np.float64)
    C0, Z0 = bench.start_factors(N, K)
    return X, C0, Z0, load_golden("converged_1500")


def _rec(X, Z, C):
    return 0.5 * np.linalg.norm(X - Z.dot(C.dot(X))) ** 2 / X.shape[0]


def _bound(fx):
    """The north star's 1e-5, or 20 x the oracle's own end-point sensitivity to a one-ulp
    perturbation of X, whichever is larger (the yardstick of tests/test_gpu_configs.py)."""
    return max(1e-5, 20.0 * float(fx["twin_rel"]))


def _run(X, C0, Z0, dtype, n_iter=T, **options):
    from convex_dim_red import _backend
    from convex_dim_red import archetypal_analysis as aa
    for name, v in options.items():
        _backend.set_option(name, v)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            Xh = X.astype(np.float32) if dtype == "float32" else X
            Z, C, _, cost, it, _, deltas = aa._iterate_aa(
                Xh, Z0, C0, np.ones(K), dtype=dtype, tolerance=0, max_iterations=n_iter,
                dictionary_solver_kwargs=dict(max_iterations=1), require_monotonic_cost_decrease=False)
    finally:
        for name in options:
            _backend.set_option(name, DEFAULTS[name])
    return Z, C, cost, it, np.asarray(deltas)


DEFAULTS = dict(qp_mode=0, qp_quad_cap=0, qp_pass_cap=24, row_local_split=1, row_local_acc64=1, proj_small=1)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_converged_parity(cdr, problem, dtype):
    """250 outer iterations: the residual-form reconstruction error (computed here in float64
    for both sides) within max(1e-5, 20 x twin) of the oracle's, every arg-max equal,
    constraints exact; the first ten costs at rounding level."""
    X, C0, Z0, fx = problem
    Z, C, cost, it, deltas = _run(X, C0, Z0, dtype)
    assert it == T - 1 and len(deltas) == T
    want = float(fx["oracle_reconstruction_error"])
    rel = abs(_rec(X, Z, C) - want) / want
    print("converged parity %s: rel diff of the reconstruction error %.3e (bound %.1e, oracle twin %.2e)"
          % (dtype, rel, _bound(fx), float(fx["twin_rel"])))
    assert rel <= _bound(fx)
    assert np.array_equal(C.argmax(axis=1), fx["oracle_argmax"])
    assert np.all(C >= 0) and np.all(Z >= 0)
    assert np.allclose(C.sum(axis=1), 1, rtol=0, atol=1e-12)
    assert np.allclose(Z.sum(axis=1), 1, rtol=0, atol=1e-12)
    # cost after each of the first ten iterations (trace form on both sides): rounding only.
    # float32 data: the trace form cancels tr(XX')/n = 369 down to the cost 60, so one
    # float32 rounding of the contractions is worth ~1e-6 of the cost.
    curve = cost - np.cumsum(deltas[::-1])[::-1] + deltas
    early = np.abs(curve[:10] - fx["cost_curve"][:10]) / fx["cost_curve"][:10]
    assert early.max() < (1e-9 if dtype == "float64" else 2e-5), early


# The alternating iteration is chaotic on this problem (two float64 runs that differ in the last
# bit are 1e-3 apart around outer iterations 60-90, where archetypes change places, and meet again
# afterwards), and it has a second local optimum, 4 of the 32 archetypes elsewhere, at
# 58.7192 = +1.0e-2: the oracle, continued for 300 iterations from a HIP end state there, stays
# (tools/diverge_continue.py).  Which optimum a run reaches is decided around iteration 80.  Of
# 80 float64 runs (oracle under data perturbations, HIP float64 under start perturbations and
# kernel choices) none went to the second optimum; of 45 float32 runs with the 32-column chains
# one did (qp_quad_cap = 1000).  So beyond the default configuration, which is pinned above, the
# families below are judged by count: every end point is one of the two optima, and at most one
# run of a family may sit in the second.
OTHER_OPTIMUM = 58.7192


def _classify(rec, fx):
    want = float(fx["oracle_reconstruction_error"])
    if abs(rec - want) / want <= _bound(fx):
        return "oracle"
    if abs(rec - OTHER_OPTIMUM) / OTHER_OPTIMUM < 1e-4:
        return "second"
    return "neither"


def test_converged_parity_from_perturbed_starts(cdr, problem):
    """The float32 run does not sit on a knife edge: of six starts that differ from bench.py's by
    1e-7 relative (float32-rounding size) at least five end at the oracle's optimum with its
    arg-max (round 2's kernel, one fp32 chain per 512 columns: 4 of 12)."""
    X, C0, Z0, fx = problem
    ends = []
    for seed in range(1, 7):
        C = C0 * (1 + 1e-7 * np.random.RandomState(100 + seed).standard_normal(C0.shape))
        C /= C.sum(axis=1, keepdims=True)
        Z, Cf, *_ = _run(X, C, Z0, "float32")
        kind = _classify(_rec(X, Z, Cf), fx)
        assert kind != "neither", seed
        if kind == "oracle":
            assert np.array_equal(Cf.argmax(axis=1), fx["oracle_argmax"]), seed
        ends.append(kind)
    print("perturbed starts, float32:", ends)
    assert ends.count("oracle") >= 5, ends


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_long_run_under_other_kernel_choices(cdr, problem, dtype):
    """The hand-over point of the four-lanes-per-sample QP kernel to the wave-per-sample kernel
    (qp_quad_cap), the other QP mappings, the column split of the row-local kernel and the
    projection's threshold search only change rounding, i.e. they re-draw the chaotic part of
    the trajectory: every variant ends at one of the two optima, at most one at the second."""
    X, C0, Z0, fx = problem
    variants = [dict(qp_quad_cap=1), dict(qp_quad_cap=8), dict(qp_quad_cap=1000), dict(qp_mode=1),
                dict(qp_mode=2), dict(qp_mode=3), dict(row_local_split=0), dict(proj_small=0)]
    ends = []
    for options in variants:
        Z, C, *_ = _run(X, C0, Z0, dtype, **options)
        kind = _classify(_rec(X, Z, C), fx)
        assert kind != "neither", options
        if kind == "oracle":
            assert np.array_equal(C.argmax(axis=1), fx["oracle_argmax"]), options
        ends.append(kind)
    print("kernel choices, %s:" % dtype, ends)
    assert ends.count("second") <= 1, list(zip(variants, ends))


@pytest.mark.parametrize("cap", [1, 3, 8, 1000])
def test_qp_quad_cap_invariance(cdr, cap):
    """Parking samples of k_qp_quad for k_qp_wave after any number of passes does not change the
    result of a weights update beyond rounding (the QpCarry hand-over), against the oracle."""
    from convex_dim_red import _backend
    from oracle import aa_oracle as orc
    rng = np.random.RandomState(11)
    n, k, p = 700, 12, 40
    W = rng.standard_normal((k, p))
    Zt = orc.right_stochastic_matrix((n, k), rng) ** 3
    Zt /= Zt.sum(axis=1, keepdims=True)
    Xs = Zt.dot(W) + 0.05 * rng.standard_normal((n, p))
    A, B = W.dot(W.T), W.dot(Xs.T)
    Z0 = orc.right_stochastic_matrix((n, k), rng)
    want, wit = orc.qp_batch(A, B, Z0, "kn", return_iters=True)
    _backend.set_option("qp_mode", 4)
    _backend.set_option("qp_quad_cap", cap)
    try:
        got, it = _backend.qp_batch(A, B, Z0, "kn", return_iters=True)
    finally:
        _backend.set_option("qp_quad_cap", 0)
        _backend.set_option("qp_mode", 0)
    assert np.abs(got - want).max() < 2e-6
    assert abs(it.mean() - wit.mean()) < 0.05 * wit.mean()
    assert np.all(got >= 0) and np.allclose(got.sum(axis=1), 1, rtol=0, atol=1e-12)


def test_qp_live_hand_over_is_bit_identical(cdr):
    """qp_live (opt-in): k_qp_quad publishes a parked sample at once and a consumer launch of the
    wave-per-sample kernel, resident on CUs of its own, continues it while k_qp_quad is still running.
    Who continues a sample does not enter its arithmetic: weights and pass counts equal the
    two-launch default bit for bit, on a QP batch and over outer iterations of a fit."""
    from convex_dim_red import _backend
    from oracle import aa_oracle as orc
    import bench
    rng = np.random.RandomState(5)
    n, k, p = 20000, 32, 64
    W = rng.standard_normal((k, p))
    Zt = orc.right_stochastic_matrix((n, k), rng) ** 3
    Zt /= Zt.sum(axis=1, keepdims=True)
    Xs = Zt.dot(W) + 0.05 * rng.standard_normal((n, p))
    A, B = W.dot(W.T), W.dot(Xs.T)
    Z0 = orc.right_stochastic_matrix((n, k), rng)
    X = bench.synthetic_rows(0, 20000)
    C0, Zs = bench.start_factors(20000, K)
    outs, traces = [], []
    try:
        _backend.set_option("qp_mode", 4)
        for live in (0, 1, 1):
            _backend.set_option("qp_live", live)
            outs.append(_backend.qp_batch(A, B, Z0, "kn", return_iters=True))
        _backend.set_option("qp_mode", 0)
        with _backend.Context(dtype="float32") as ctx:
            ctx.set_data(X)
            for live in (0, 1):
                _backend.set_option("qp_live", live)
                ctx.set_state(C0, Zs, np.ones(K))
                ctx.prepare()
                traces.append(np.asarray(ctx.outer_iterations(12, dict(max_iterations=1), {})))
    finally:
        _backend.set_option("qp_live", 0)
        _backend.set_option("qp_mode", 0)
    assert outs[0][1].max() > 24                      # samples were handed over
    for got, it in outs[1:]:
        assert np.array_equal(got, outs[0][0]) and np.array_equal(it, outs[0][1])
    assert np.array_equal(traces[0], traces[1])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_schedule_knobs_do_not_change_a_bit(dtype):
    """Round 4 moved work around without touching arithmetic: the sample order of the next weights update is
    formed inside the previous continuation launch (qp_fused_order), the wave-per-sample and four-lane kernels
    decide their stopping tests lazily behind a projected-gradient bound (qp_wave_lazy, qp_quad_lazy: the same
    decisions), the projections' reductions are finalized by the last block of their pass in the fixed order of
    the finalize kernel (fin_in_last), the dictionary set-up rides in the gradient launch (setup_in_grad) and
    Z'Z may run on the side stream (gram_side).  Each of them off / on: costs, factors and the QP's pass counts
    of a 40 000-row fit (the pass kernels of the timed configuration, samples parked at the pass cap) are equal
    bit for bit."""
    from convex_dim_red import _backend
    import bench
    n = 40000
    X = bench.synthetic_rows(0, n)
    C0, Z0 = bench.start_factors(n, K)
    knobs = [("qp_fused_order", 0), ("qp_wave_lazy", 0), ("qp_quad_lazy", 1), ("fin_in_last", 0),
             ("setup_in_grad", 0), ("gram_side", 1)]
    defaults = {"qp_fused_order": 1, "qp_wave_lazy": 1, "qp_quad_lazy": 0, "fin_in_last": 1, "setup_in_grad": 1,
                "gram_side": 0}

    def run():
        with _backend.Context(dtype=dtype) as ctx:
            ctx.set_data(X)
            ctx.set_state(C0, Z0, np.ones(K))
            ctx.prepare()
            costs = np.asarray(ctx.outer_iterations(7, dict(max_iterations=1), {}))
            C, Z, _ = ctx.get_state()
            return costs, C, Z

    base = run()
    assert np.all(np.diff(base[0]) < 0)
    try:
        for name, value in knobs:
            _backend.set_option(name, value)
            got = run()
            _backend.set_option(name, defaults[name])
            for a, b in zip(base, got):
                assert np.array_equal(a, b), name
    finally:
        for name, value in defaults.items():
            _backend.set_option(name, value)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("n", [1500, 40000])
def test_pass_kernels_against_numpy(cdr, dtype, n):
    """The two contractions every update is built from, on their own, against float64 NumPy
    (archetypal_analysis.py:614-615,627-628): errors relative to sum |x||b|, the scale rounding
    errors live on.  float64: 2e-14 (4096-term float64 chains).  float32 (X exact in float32, operand rounded to
    float32, fp32 products): the row-local kernel below 32 768 rows sums 32-column pieces in
    float64 (<= 5e-8 of the yardstick, measured 2.3e-8); the wave-streaming kernel of larger
    shards keeps one fp32 chain per row (<= 6e-6, measured 2.7e-6)."""
    import bench
    from convex_dim_red import _backend
    X32 = bench.synthetic_rows(0, n)
    X = X32.astype(np.float64)
    rng = np.random.RandomState(3)
    C = rng.uniform(size=(K, n))
    C /= C.sum(axis=1, keepdims=True)
    B = C.dot(X)                                         # archetype-like rows: the coherent case
    A = rng.uniform(size=(n, K))
    A /= A.sum(axis=1, keepdims=True)
    with _backend.Context(dtype=dtype) as ctx:
        ctx.set_data(X32 if dtype == "float32" else X)
        got_rl = ctx.pass_row_local(B)
        got_rr = ctx.pass_reduce_rows(A)
    err_rl = (np.abs(got_rl - X.dot(B.T)) / np.abs(X).dot(np.abs(B).T)).max()
    err_rr = (np.abs(got_rr - A.T.dot(X)) / np.abs(A).T.dot(np.abs(X))).max()
    print("pass kernels %s n=%d: row-local %.2e, reduce-rows %.2e of sum|x||b|" % (dtype, n, err_rl, err_rr))
    if dtype == "float64":
        assert err_rl < 2e-14 and err_rr < 2e-14
    else:
        assert err_rl < 5e-8
        assert err_rr < 2e-7
