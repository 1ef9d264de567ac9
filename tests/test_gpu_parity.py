"""Parity of the HIP path (through the C ABI, via the product package) against the
reference-generated golden vectors and the CPU oracle.  Needs an MI355X.

Bars: support patterns / argmax indices / constraint satisfaction exact; values of
fixed-iteration runs to rounding (<= 1e-9, stated per test); runs to convergence to
the solver's own stopping tolerance; float32-MFMA mode to the stated fp32 tolerance."""
import warnings

import numpy as np
import pytest

from conftest import f32_perturbed, load_golden, oracle_twins, ulp_perturbed

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cdr():
    import convex_dim_red
    from convex_dim_red import _backend
    _backend.require_gpu()
    return convex_dim_red


@pytest.fixture(scope="module")
def orc():
    from oracle import aa_oracle
    return aa_oracle


def _keys(g, prefix):
    return sorted(k[len(prefix):] for k in g.files if k.startswith(prefix))


def _assert_simplex(M, atol=1e-13):
    assert np.all(M >= 0)
    assert np.allclose(M.sum(axis=1), 1, rtol=0, atol=atol)


# ---------------------------------------------------------------- simplex projection
def test_simplex_rows_golden(cdr):
    g = load_golden("simplex_rows")
    for key in _keys(g, "in_"):
        A, want = g["in_" + key], g["out_" + key]
        got = cdr.simplex_project_rows(A)
        scale = max(1.0, np.abs(A).max())
        assert np.allclose(got, want, rtol=0, atol=8e-16 * scale * max(1, A.shape[1]) ** 0.5), key
        if key not in ("ties", "large"):
            assert np.array_equal(got > 0, want > 0), key       # support pattern exact
        cols = cdr.simplex_project_columns(np.ascontiguousarray(A.T))
        assert np.array_equal(cols, got.T), key


def test_simplex_exact_small_cases(cdr):
    # reference tests/test_simplex_projection.py:13-57,166-176 known answers
    assert np.array_equal(cdr.simplex_project_rows(np.array([[-0.5]])), [[1.0]])
    got = cdr.simplex_project_rows(np.array([[0.8, 0.8], [0.0, 2.0], [0.5, -0.5]]))
    assert np.array_equal(got, [[0.5, 0.5], [0.0, 1.0], [1.0, 0.0]])
    got = cdr.simplex_project_rows(np.array([[0.5, 0.5], [0.5, 1.0], [0.0, -0.5]]))
    assert np.array_equal(got, [[0.5, 0.5], [0.25, 0.75], [0.75, 0.25]])


@pytest.mark.parametrize("shape", [(57, 5), (341, 317), (7, 3000), (2, 20000), (3, 1)])
def test_simplex_rows_vs_oracle(cdr, orc, shape):
    rng = np.random.RandomState(shape[0] * 7 + shape[1])
    A = rng.standard_normal(shape) * rng.choice([0.01, 1.0, 50.0])
    got, want = cdr.simplex_project_rows(A), orc.simplex_project_rows(A)
    assert np.abs(got - want).max() < 1e-13 * max(1.0, np.abs(A).max())
    assert np.array_equal(got > 0, want > 0)
    _assert_simplex(got, 1e-12)
    inside = orc.right_stochastic_matrix(shape, rng)            # already feasible: invariant
    assert np.abs(cdr.simplex_project_rows(inside) - inside).max() < 1e-15


# ---------------------------------------------------------------- per-sample QP
@pytest.fixture(params=[1, 2, 3, 4], ids=["wave-per-sample", "lane+wave", "row", "quad"])
def qp_kernel(request):
    """All four mappings of the batched QP (default: four lanes per sample for k <= 32, one wave
    per sample above)."""
    from convex_dim_red import _backend
    _backend.set_option("qp_mode", request.param)
    yield request.param
    _backend.set_option("qp_mode", 0)


@pytest.mark.parametrize("k", [3, 8, 10, 32])
@pytest.mark.parametrize("tag,kw", [("default", {}), ("one", dict(max_iterations=1)),
                                    ("alpha0", dict(alpha0=0.5, max_iterations=5))])
def test_qp_golden(cdr, qp_kernel, k, tag, kw):
    from convex_dim_red import _backend
    g = load_golden("quad_simplex_spg")
    A, B, Z0 = g["in_A_k%d" % k], g["in_B_k%d" % k], g["in_Z0_k%d" % k]
    want = g["out_Z_k%d_%s" % (k, tag)]
    got = _backend.qp_batch(A, B, Z0, "kn", **kw)
    scale = np.abs(A).max()
    tol = 1e-6 if tag == "default" else 1e-11 * scale
    assert np.abs(got - want).max() < tol
    _assert_simplex(got)
    if tag != "default":
        assert np.array_equal(got > 0, want > 0)
    # objective value agrees much more tightly than the minimiser
    f = lambda Z: 0.5 * np.einsum("ti,ij,tj->t", Z, A, Z) - np.einsum("ti,it->t", Z, B)
    assert np.abs(f(got) - f(want)).max() < 1e-9 * scale


def test_qp_layouts_and_iters(cdr, orc, qp_kernel):
    from convex_dim_red import _backend
    rng = np.random.RandomState(5)
    n, k, p = 500, 10, 30
    W = rng.standard_normal((p, k))
    X = orc.right_stochastic_matrix((n, k), rng).dot(W.T) + 0.1 * rng.standard_normal((n, p))
    A, XW = W.T.dot(W), X.dot(W)
    Z0 = orc.right_stochastic_matrix((n, k), rng)
    got_nk, it = _backend.qp_batch(A, XW, Z0, "nk", return_iters=True)
    got_kn = _backend.qp_batch(A, np.ascontiguousarray(XW.T), Z0, "kn")
    want, it_o = orc.qp_batch(A, XW, Z0, "nk", return_iters=True)
    assert np.array_equal(got_nk, got_kn)
    assert np.abs(got_nk - want).max() < 1e-6
    assert abs(it.mean() - it_o.mean()) < 0.05 * it_o.mean() + 1
    from convex_dim_red.spg import quad_simplex_spg
    single = quad_simplex_spg(A, -XW[3], Z0[3])
    assert np.abs(single - want[3]).max() < 1e-6


# ---------------------------------------------------------------- dictionary SPG
def test_dictionary_spg_golden(cdr):
    from convex_dim_red import archetypal_analysis as aa
    g = load_golden("aa_dictionary_spg")
    X, C0, Z0 = g["in_X"], g["in_C0"], g["in_Z0"]
    alpha = np.ones(C0.shape[0])
    trX = np.trace(X.dot(X.T))
    XXtZ = X.dot(X.T.dot(Z0))
    ZtZ = Z0.T.dot(Z0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag, kw, tol in (("one", dict(max_iterations=1), 1e-11),
                             ("five", dict(max_iterations=5), 1e-9)):
            C = aa._update_aa_dictionary(X, C0, alpha, trX, XXtZ, ZtZ, **kw)
            assert np.abs(C - g["out_C_" + tag]).max() < tol, tag
            assert np.array_equal(C > 0, g["out_C_" + tag] > 0), tag
            _assert_simplex(C)
        K = X.dot(X.T)
        C = aa._update_kernel_aa_dictionary(K, C0, alpha, np.trace(K), K.dot(Z0), ZtZ,
                                            max_iterations=1)
        assert np.abs(C - g["out_kernel_C_one"]).max() < 1e-11
        CK = C0.dot(K)
        Z = aa._update_kernel_aa_weights(Z0, alpha, CK, CK.dot(C0.T))
        assert np.abs(Z - g["out_kernel_Z"]).max() < 1e-6
        assert abs(aa._kernel_aa_cost(K, Z0, C0, alpha) - g["out_kernel_cost"]) < 1e-12


def test_dictionary_spg_flags_and_stats(cdr):
    from convex_dim_red import _backend
    g = load_golden("aa_dictionary_spg")
    X, C0, Z0 = g["in_X"], g["in_C0"], g["in_Z0"]
    for tag, kw in (("one", dict(max_iterations=1)), ("five", dict(max_iterations=5))):
        want_f, want_it, want_fe = g["out_stats_" + tag]
        with _backend.Context(dtype="float64") as ctx:
            ctx.set_data(X)
            ctx.set_state(C0, Z0, np.ones(C0.shape[0]))
            ctx.prepare()
            st = ctx.dictionary_update(**kw)
        assert abs(st.f - want_f) < 1e-11 * abs(want_f)
        assert (st.n_iter, st.n_feval) == (int(want_it), int(want_fe))
        assert st.flags & _backend.SPG_FLAG_MAX_ITER


# ---------------------------------------------------------------- outer loops
def test_iterate_aa_steps_golden(cdr):
    from convex_dim_red import archetypal_analysis as aa
    g = load_golden("iterate_aa")
    X, C0, Z0 = g["in_X"], g["in_C0"], g["in_Z0"]
    k = C0.shape[0]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Z, C = Z0.copy(), C0.copy()
        for step in range(3):
            Z, C, _, cost, n_iter, _, deltas = aa._iterate_aa(
                X, Z, C, np.ones(k), tolerance=0, max_iterations=1,
                dictionary_solver_kwargs=dict(max_iterations=1),
                require_monotonic_cost_decrease=False)
            assert np.abs(C - g["out_step%d_C" % step]).max() < 1e-8, step
            assert np.abs(Z - g["out_step%d_Z" % step]).max() < 5e-6, step
            assert abs(cost - g["out_step%d_cost" % step]) < 1e-9, step
            assert n_iter == 0 and len(deltas) == 1
            _assert_simplex(C)
            _assert_simplex(Z)


_TWIN_CACHE = {}


def _abs_stop_window(ref_deltas, threshold=1e-6, factor=1.5):
    """`_stop_window` for the |delta cost| < threshold rule (archetypal_analysis.py:177-197)."""
    change = np.abs(np.asarray(ref_deltas))
    big = change >= factor * threshold
    return int(np.argmax(big[::-1])) if np.any(big) else len(change)


def _trace_noise(X, dtype):
    """float32 mode: noise of one evaluation of the trace-form cost, which cancels tr(XX')/n down
    to the residual (the monotonicity check of the float32 mode allows for the
    same amount): 8 eps32 tr(XX')/n.  Zero for float64 data."""
    return 0.0 if dtype == "float64" else 8 * 6e-8 * float((np.asarray(X, dtype=np.float64) ** 2).sum()) / X.shape[0]


def _stop_range(ref_deltas, want_it, noise, t_shift, threshold=1e-6):
    """(earliest, latest) outer iteration at which a run may fire |delta cost| < threshold when its
    cost changes are the reference's up to `noise` per evaluation (float32: `_trace_noise`; the rule
    then fires by chance as soon as the true change is of the size of the noise) and up to last-bit
    differences (window in which the reference's own changes stay within 1.5 x of the threshold,
    twice over, and twice what the oracle's perturbed twins move by)."""
    change = np.abs(np.asarray(ref_deltas))
    small = np.nonzero(change < 1.5 * threshold + 2 * noise)[0]
    earliest = int(small[0]) if len(small) else int(want_it)
    slack = max(2 * _abs_stop_window(ref_deltas, threshold), 2 * t_shift)
    return min(earliest, int(want_it) - slack), int(want_it) + slack


def _cost_gap(ref_deltas, n_iter, want_it, threshold=1e-6):
    """What the end cost may differ by when the run stopped at n_iter instead of want_it: the
    reference's own cost changes of the iterations it did not run, or 1.5 x threshold for every
    iteration it ran longer."""
    change = np.abs(np.asarray(ref_deltas))
    if n_iter < want_it:
        return float(change[n_iter + 1:int(want_it) + 1].sum())
    return (n_iter - int(want_it)) * 1.5 * threshold


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_iterate_aa_traces_golden(cdr, orc, qp_kernel, dtype):
    """_iterate_aa to the |delta cost| < 1e-6 rule against the reference's runs (production and
    default dictionary solver, and delta = 0.1 with scale factors).  Yardsticks, all computed:
    the oracle (bit-identical to the reference on these runs: tests/test_oracle_golden.py) is run
    again on the data moved by one ulp (float64 legs) or by one float32 rounding (float32 legs:
    three draws); the HIP path has to agree with the reference within 20 x what those twins do to
    the oracle's own cost, stopping iteration and scale factors (floor: 1e-9 relative, the
    rounding of a 30-iteration run) -- plus, where the stopping iteration differs, the reference's
    own cost changes of the iterations in between (`_cost_gap`) and one iteration's movement of the
    scale factors per iteration of shift; float32 adds the noise of the trace-form cost
    (`_trace_noise`, 5e-6 here against a threshold of 1e-6: the float32 mode cannot resolve this
    stopping rule on nearly noise-free data and may fire from the iteration on at which the
    reference's changes drop to the size of that noise, `_stop_range`).  float64: the stopping
    iteration is the reference's, exactly, on the delta = 0 legs."""
    from convex_dim_red import archetypal_analysis as aa
    g = load_golden("iterate_aa")
    X, C0, Z0 = g["in_X"], g["in_C0"], g["in_Z0"]
    Xh = X.astype(np.float32) if dtype == "float32" else X
    k = C0.shape[0]
    noise = _trace_noise(X, dtype)

    def oracle(Xin, alpha0, **kw):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return orc.iterate_aa(Xin, Z0.copy(), C0.copy(), alpha0.copy(), **dict(dict(tolerance=1e-6), **kw))

    def twin_runs(alpha0, **kw):
        key = (dtype, repr(sorted(kw.items())), alpha0.tobytes())
        if key not in _TWIN_CACHE:              # the same for every QP mapping this test is run with
            one_pass = kw.get("dictionary_solver_kwargs", {}).get("max_iterations", 1000) == 1
            _TWIN_CACHE[key] = oracle_twins(orc, lambda Xin: oracle(Xin, alpha0, **kw), X, dtype, operands=one_pass)
        return _TWIN_CACHE[key]

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag, kw in (("prod", dict(dictionary_solver_kwargs=dict(max_iterations=1))),
                        ("default", {})):
            Z, C, al, cost, n_iter, _, deltas = aa._iterate_aa(
                Xh, Z0.copy(), C0.copy(), np.ones(k), tolerance=1e-6, max_iterations=60,
                dtype=dtype, **kw)
            want_cost, want_it = g["out_cost_" + tag]
            twins = twin_runs(np.ones(k), max_iterations=60, **kw)
            t_cost = max(abs(t[3] - want_cost) for t in twins)
            t_shift = max(abs(t[4] - int(want_it)) for t in twins)
            window = _abs_stop_window(g["out_deltas_" + tag])
            shift = abs(n_iter - int(want_it))
            lo, hi = _stop_range(g["out_deltas_" + tag], want_it, noise, t_shift)
            tol = max(1e-9 * want_cost, 20 * t_cost) + _cost_gap(g["out_deltas_" + tag], n_iter, want_it) + noise
            print("iterate_aa %s %s: n_iter %d (reference %d, may stop in %d..%d, twins move by %d), |dcost| %.2e (bound %.2e)"
                  % (dtype, tag, n_iter, int(want_it), lo, hi, t_shift, abs(cost - want_cost), tol))
            if dtype == "float64":
                assert n_iter == int(want_it), tag
            else:
                assert lo <= n_iter <= hi, (tag, n_iter, lo, hi)
            assert abs(cost - want_cost) <= tol, (tag, abs(cost - want_cost), tol)
            assert np.array_equal(C.argmax(axis=1), g["out_C_" + tag].argmax(axis=1)), tag
            assert len(deltas) == n_iter + 1
        al0 = g["in_alpha0"]
        dkw = dict(delta=0.1, dictionary_solver_kwargs=dict(max_iterations=1))
        Z, C, al, cost, n_iter, _, deltas = aa._iterate_aa(
            Xh, Z0.copy(), C0.copy(), al0.copy(), tolerance=1e-6, max_iterations=40, dtype=dtype, **dkw)
        want_cost, want_it = g["out_cost_delta"]
        twins = twin_runs(al0, max_iterations=40, **dkw)
        t_cost = max(abs(t[3] - want_cost) for t in twins)
        t_alpha = max(np.abs(t[2] - g["out_alpha_delta"]).max() for t in twins)
        t_shift = max(abs(t[4] - int(want_it)) for t in twins)
        # one iteration's movement of the scale factors at the reference's stop (alpha converges slowly)
        prev = oracle(X, al0, max_iterations=int(want_it), **dict(dkw, tolerance=0))[2]
        a_step = np.abs(prev - g["out_alpha_delta"]).max()
        window = _abs_stop_window(g["out_deltas_delta"])
        shift = abs(n_iter - int(want_it))
        print("iterate_aa %s delta: n_iter %d (reference %d, window %d, twins move by %d), |dcost| %.2e, "
              "max |dalpha| %.2e (twins %.2e, one iteration %.2e)"
              % (dtype, n_iter, int(want_it), window, t_shift, abs(cost - want_cost),
                 np.abs(al - g["out_alpha_delta"]).max(), t_alpha, a_step))
        lo, hi = _stop_range(g["out_deltas_delta"], want_it, noise, t_shift)
        assert lo <= n_iter <= hi, (n_iter, lo, hi)
        assert abs(cost - want_cost) <= (max(1e-9 * want_cost, 20 * t_cost)
                                         + _cost_gap(g["out_deltas_delta"], n_iter, want_it) + noise)
        assert np.abs(al - g["out_alpha_delta"]).max() <= max(1e-10, 20 * t_alpha) + 2 * shift * a_step


def test_iterate_kernel_aa_golden(cdr):
    from convex_dim_red import archetypal_analysis as aa
    g = load_golden("iterate_kernel_aa")
    K, C0, Z0 = g["in_K"], g["in_C0"], g["in_Z0"]
    k = C0.shape[0]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag, kw in (("prod", dict(dictionary_solver_kwargs=dict(max_iterations=1))),
                        ("default", {})):
            Z, C, al, cost, n_iter, _, deltas = aa._iterate_kernel_aa(
                K, Z0.copy(), C0.copy(), np.ones(k), tolerance=1e-6, max_iterations=40, **kw)
            want_cost, want_it = g["out_cost_" + tag]
            assert abs(cost - want_cost) < 2e-6, tag
            assert n_iter == int(want_it), tag
            _assert_simplex(C, 1e-12)
            _assert_simplex(Z, 1e-12)


# ---------------------------------------------------------------- estimators
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_aa_estimator_known_answers(cdr, orc, dtype):
    """ArchetypalAnalysis.fit_transform to |delta cost| < 1e-6 against the reference's four known
    answers.  float64: stopping iteration exact, cost within 20 x the oracle's one-ulp
    twin (floor 1e-9 relative).  float32: the trace-form cost carries `_trace_noise` = 7e-6 of noise
    per evaluation on these nearly noise-free data, seven times the stopping threshold, so the rule
    fires by chance from the iteration on at which the reference's own changes reach that size
    (`_stop_range`, from the oracle's trace of the same run); the end cost then lacks exactly the
    reference's changes of the iterations not run (`_cost_gap`), plus that noise, plus 20 x what
    float32-sized perturbations of the data do to the oracle's end cost."""
    g = load_golden("aa_estimator")
    X = g["in_X"]
    Xh = X.astype(np.float32) if dtype == "float32" else X
    noise = _trace_noise(X, dtype)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for init in ("furthest_sum", "random"):
            for tag, dkw in (("one", dict(max_iterations=1)), ("full", {})):
                key = "%s_%s" % (init, tag)
                kw = dict(init=init, random_state=0, tolerance=1e-6, max_iterations=1000,
                          dictionary_solver_kwargs=dkw)
                m = cdr.ArchetypalAnalysis(3, dtype=dtype, **kw)
                W = m.fit_transform(Xh)
                want_cost, want_it = g["out_cost_" + key]
                base = orc.archetypal_analysis(X, 3, **kw)
                assert base["n_iter"] == int(want_it) and abs(base["cost"] - want_cost) < 1e-8
                twins = oracle_twins(orc, lambda Xin: orc.archetypal_analysis(Xin, 3, **kw), X, dtype,
                                     operands=(tag == "one"))
                t_cost = max(abs(t["cost"] - want_cost) for t in twins)
                t_shift = max(abs(t["n_iter"] - int(want_it)) for t in twins)
                lo, hi = _stop_range(base["cost_deltas"], want_it, noise, t_shift)
                tol = (max(1e-9 * want_cost, 20 * t_cost) + _cost_gap(base["cost_deltas"], m.n_iter, want_it)
                       + noise)
                print("aa estimator %s %s: n_iter %d (reference %d, may stop in %d..%d), |dcost| %.2e (bound %.2e)"
                      % (dtype, key, m.n_iter, int(want_it), lo, hi, abs(m.cost - want_cost), tol))
                if dtype == "float64":
                    assert m.n_iter == int(want_it), key
                else:
                    assert lo <= m.n_iter <= hi, (key, m.n_iter, lo, hi)
                assert abs(m.cost - want_cost) <= tol, (key, abs(m.cost - want_cost), tol)
                assert sorted(m.dictionary.argmax(axis=1)) == sorted(g["out_argmax_" + key]), key
                # archetypes = dictionary . data in the arithmetic of the data; float32: rounding of X plus
                # an fp32 accumulation chain, <= 16 eps32 sum |c||x| <= 16 eps32 max |x| (rows of C sum to 1)
                atol = 1e-12 if dtype == "float64" else 16 * 6e-8 * np.abs(X).max()
                assert np.abs(m.archetypes - m.dictionary.dot(X)).max() < atol
                _assert_simplex(W, 1e-12)
                if key == "furthest_sum_one" and dtype == "float64":
                    Wn, cn = m.transform(X[:25] + 0.0)
                    # same generator state as the reference's after its fit, so the same fresh
                    # starting weights: the weights themselves are comparable, not only the cost
                    # (weights-only loop to the 1e-6 rule: within 10 x that tolerance)
                    assert abs(cn - g["out_transform_cost"]) < 1e-8
                    assert np.abs(Wn - g["out_transform_W"]).max() < 1e-5
                    assert np.abs(m.inverse_transform(Wn) - g["out_inverse"]).max() < 1e-5
                    _assert_simplex(Wn, 1e-12)


def test_kernel_aa_estimator_hull(cdr):
    g = load_golden("kernel_aa_estimator")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = cdr.KernelAA(3, delta=0, init="custom", tolerance=1e-10, max_iterations=500)
        W = m.fit_transform(g["in_K"], dictionary=g["in_C0"], weights=g["in_Z0"], alpha=np.ones(3))
    assert sorted(m.dictionary.argmax(axis=1)) == [5, 27, 32]
    assert abs(m.cost - g["out_cost"][0]) < 1e-8
    _assert_simplex(W, 1e-12)


def test_deepcopy_and_shared_random_state(cdr):
    import copy
    rng = np.random.RandomState(3)
    X = rng.uniform(size=(80, 12))
    shared = np.random.RandomState(0)
    best = None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(2):                        # drivers' n_init loop, bin/run_hadisst_aa.py:158-172
            m = cdr.ArchetypalAnalysis(3, init="random", random_state=shared, max_iterations=20,
                                       dictionary_solver_kwargs=dict(max_iterations=1))
            m.fit_transform(X)
            if best is None or m.cost < best.cost:
                best = copy.deepcopy(m)
    assert best.weights.shape == (80, 3) and best.archetypes.shape == (3, 12)


# ---------------------------------------------------------------- GPNH
def test_gpnh_golden(cdr, qp_kernel):
    from convex_dim_red import gpnh_convex_coding as gp
    g = load_golden("gpnh")
    X, W0, Z0 = g["in_X"], g["in_W0"], g["in_Z0"]
    p, k = W0.shape
    for lam in (0.0, 1.0):
        tag = "lam%d" % int(lam)
        assert abs(gp._gpnh_cost(X, Z0, W0, lam) - g["out_cost0_" + tag]) < 1e-12
        GW = (4.0 / (p * k * (k - 1))) * (k * np.eye(k) - 1)
        Wn = gp._update_gpnh_dictionary(X, Z0, Z0.T.dot(Z0), GW, lam)
        assert np.abs(Wn - g["out_Wupd_" + tag]).max() < 1e-10
        for wtag, wkw in (("one", dict(max_iterations=1)), ("full", {})):
            Z, W, cost, n_iter, _, deltas = gp._iterate_gpnh_convex_coding(
                X, Z0.copy(), W0.copy(), lambda_W=lam, tolerance=1e-6, max_iterations=200,
                stopping_criterion="rel_delta_f", weights_solver_kwargs=wkw)
            want_cost, want_it = g["out_cost_%s_%s" % (tag, wtag)]
            want_deltas = g["out_deltas_%s_%s" % (tag, wtag)]
            # fixed-iteration part: the cost change of each of the first ten outer iterations at
            # rounding level (cost ~ 1, changes down to 1e-4)
            m = min(10, len(deltas), len(want_deltas))
            assert np.abs(np.asarray(deltas[:m]) - want_deltas[:m]).max() < 1e-10, (tag, wtag)
            # run to the stopping rule |delta cost| / cost < 1e-6.  Yardstick from the reference's
            # own trace: near the end its relative cost change hovers within a factor 1.5 of the
            # threshold for `window` iterations (with one SPG pass per outer iteration it is not
            # even monotone there: 1.24, 1.20, 1.35, 1.19, 1.03, 0.98 e-6), so a run whose changes
            # differ in the 8th digit may fire anywhere in that window, before or after, and its
            # end cost then differs by the changes in between, each < 1.5e-6 of the cost.  With the
            # full QP the trace is smooth and the stop is the reference's (asserted exactly).
            window = _stop_window(want_deltas, want_cost)
            shift = abs(n_iter - int(want_it))
            print("gpnh golden %s %s: n_iter %d (reference %d, window %d), cost rel diff %.2e"
                  % (tag, wtag, n_iter, int(want_it), window, abs(cost - want_cost) / want_cost))
            if wtag == "full":
                assert shift == 0, (tag, wtag)
                assert abs(cost - want_cost) < 1e-10 * want_cost, (tag, wtag)
            else:
                assert shift <= 2 * window, (tag, wtag, window)
                assert abs(cost - want_cost) <= (shift + 1) * 1.5e-6 * want_cost, (tag, wtag)
            _assert_simplex(Z, 1e-12)
    assert np.abs(gp._update_gpnh_weights(X, Z0, W0) - g["out_Zupd"]).max() < 1e-6


def _stop_window(ref_deltas, ref_cost, threshold=1e-6, factor=1.5):
    """Trailing iterations of the reference's own trace whose relative cost change stays within
    `factor` of the stopping threshold: a run whose changes differ from the reference's in the 8th
    digit may fire the |delta cost| / cost < threshold rule anywhere in a window of that length
    before or after the reference's stop (with one SPG pass per outer iteration the changes are
    not even monotone there), and its end cost differs by the changes in between."""
    rel_change = np.abs(np.asarray(ref_deltas)) / ref_cost
    big = rel_change >= factor * threshold
    return int(np.argmax(big[::-1])) if np.any(big) else len(rel_change)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_gpnh_estimator_known_answers(cdr, dtype):
    """GPNHConvexCoding.fit_transform to the stopping rule (production setting: one SPG pass per
    weights update, rel_delta_f at 1e-6) against the reference's runs.  float64: the first ten
    cost changes at rounding level, the stopping iteration inside the reference's own window
    (`_stop_window`), the end cost within the changes of the iterations in between.  float32
    data rounds X itself (6e-8 relative): the cost level moves by ~1e-6 relative, the yardstick
    for the stop is the same window."""
    g = load_golden("gpnh_estimator")
    X = g["in_X"]
    for lam in (0.0, 1.0):
        for init in ("random", "furthest_sum"):
            key = "lam%d_%s" % (int(lam), init)
            m = cdr.GPNHConvexCoding(5, lambda_W=lam, init=init, tolerance=1e-6,
                                     max_iterations=3000, stopping_criterion="rel_delta_f",
                                     random_state=0, dtype=dtype,
                                     weights_solver_kwargs=dict(max_iterations=1))
            Z = m.fit_transform(X)
            want_cost, want_it = g["out_cost_" + key]
            want_deltas = g["out_deltas_" + key]
            window = _stop_window(want_deltas, want_cost)
            shift = abs(m.n_iter - int(want_it))
            print("gpnh estimator %s %s: n_iter %d (reference %d, window %d), cost rel diff %.2e"
                  % (dtype, key, m.n_iter, int(want_it), window, abs(m.cost - want_cost) / want_cost))
            early = np.abs(np.asarray(m.cost_deltas[:10]) - want_deltas[:10]).max()
            assert early < (1e-10 if dtype == "float64" else 2e-5), key
            assert shift <= 2 * window, (key, window)
            assert abs(m.cost - want_cost) <= ((shift + 1) * 1.5e-6 + (0 if dtype == "float64" else 5e-6)) * want_cost, key
            assert m.dictionary.shape == (30, 5) and Z.shape == (300, 5)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_gpnh_transform_golden(cdr, orc, dtype):
    """GPNHConvexCoding.transform / inverse_transform (gpnh_convex_coding.py:623-668) against the
    reference: the reference's fitted dictionary, the generator in its state after that fit
    (randn(p, k) and uniform(n, k) drawn), fresh random weights, weights-only loop on the device.
    Every (lambda_W, QP setting, dtype) combination is checked -- none is skipped -- in two tiers:

    * FIXED 1 and 3 weights-only iterations from the reference's own start (`out_start_*`, the
      draw the transform makes) against the reference's weights after as many iterations:
      float64 at rounding level (one SPG pass per update) / at the QP's stopping tolerance (full
      QP); float32 data within 20 x what the ORACLE's weights move under float32-sized
      perturbations (`conftest.oracle_twins`: the data moved by 6e-8 relative, and the operands of the big
      contractions rounded to float32 as the float32 mode feeds them to the matrix cores);
    * the run to the stopping rule |delta cost| / cost < 1e-6 (`transform` itself): with one pass
      per update the reference's relative cost changes near its stop read 3.0, 3.9, 3.2, 1.7, 2.2,
      0.72 e-6 -- not monotone -- so the iteration at which a run with other last bits fires is
      judged by the reference's own window (`_stop_window`) and by how far the oracle's stop moves
      under perturbations (one ulp for the float64 legs, float32-sized for the float32 legs); the
      cost may then differ by the reference's own changes of the iterations not run (`_cost_gap`)
      plus 20 x what those perturbations do to the oracle's end cost at its own stop plus, for
      float32 data, 20 x the change of the cost at the reference's end point when the data are
      rounded to float32 and the noise of the trace-form cost (`_trace_noise`); the weights by
      what the reference's own weights still move in one iteration at its stop
      (`out_Wprev_*`: 9e-4 / 2.9e-3 with one pass, 5e-8 / 8e-7 with the full QP) per iteration of
      shift, or 10 x the QP tolerance."""
    from convex_dim_red import gpnh_convex_coding as gp
    g = load_golden("gpnh_transform")
    X, Xn = g["in_X"], g["in_Xnew"]
    Xn_h = Xn.astype(np.float32) if dtype == "float32" else Xn
    Xn_r = Xn_h.astype(np.float64)                       # the data the device holds, as float64
    for lam in (0.0, 1.0):
        for wtag, wkw in (("one", dict(max_iterations=1)), ("full", {})):
            key = "lam%d_%s" % (int(lam), wtag)
            W = g["out_dictionary_" + key]
            Z0 = g["out_start_" + key]
            run = dict(lambda_W=lam, update_dictionary=False, update_weights=True,
                       stopping_criterion="rel_delta_f", weights_solver_kwargs=wkw)

            def oracle(Xin, **kw):
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    # (no monotonicity check: a perturbed twin may step back by its own noise)
                    return orc.iterate_gpnh(Xin, Z0.copy(), W.copy(),
                                            **dict(run, require_monotonic_cost_decrease=False, **kw))

            # ---- fixed iterations
            for iters in (1, 3):
                want = g["out_W%d_%s" % (iters, key)]
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    got = gp._iterate_gpnh_convex_coding(Xn_h, Z0.copy(), W.copy(), dtype=dtype, tolerance=0,
                                                         max_iterations=iters, **run)[0]
                floor = 1e-10 if wtag == "one" else 1e-6          # rounding / the QP's stopping tolerance
                tol = floor
                if dtype == "float32":
                    base = oracle(Xn, tolerance=0, max_iterations=iters)[0]
                    twins = oracle_twins(orc, lambda Xin: oracle(Xin, tolerance=0, max_iterations=iters)[0], Xn, dtype)
                    tol = max(floor, 20 * max(np.abs(t - base).max() for t in twins))
                err = np.abs(got - want).max()
                print("gpnh transform %s %s: %d iteration(s) max |dW| %.2e (bound %.2e)" % (dtype, key, iters, err, tol))
                assert err <= tol, (key, iters, err, tol)
                _assert_simplex(got, 1e-12)
            # ---- to the stopping rule, through transform()
            rs = np.random.RandomState(0)
            rs.randn(X.shape[1], 5)
            rs.uniform(size=(X.shape[0], 5))
            m = cdr.GPNHConvexCoding(5, lambda_W=lam, init="random", tolerance=1e-6, max_iterations=400,
                                     stopping_criterion="rel_delta_f", random_state=rs, dtype=dtype,
                                     weights_solver_kwargs=wkw)
            m.dictionary = W.copy()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                Wn, cn = m.transform(Xn_h)
                direct = gp._iterate_gpnh_convex_coding(Xn_h, Z0.copy(), W.copy(), dtype=dtype, tolerance=1e-6,
                                                        max_iterations=400, **run)
            assert np.array_equal(Wn, direct[0]) and cn == direct[2]      # transform() is that loop from that start
            n_iter = direct[3]
            want_cost, want_it = g["out_trace_" + key]
            assert want_cost == g["out_cost_" + key][0]
            ref_deltas = g["out_deltas_" + key]
            twins = oracle_twins(orc, lambda Xin: oracle(Xin, tolerance=1e-6, max_iterations=400), Xn, dtype)
            t_shift = max(abs(t[3] - int(want_it)) for t in twins)
            # what the perturbation does to the end point beyond moving the stop: twins compared at their
            # own stopping iteration against the reference's trace at that iteration
            curve = want_cost - np.cumsum(ref_deltas[::-1])[::-1] + ref_deltas       # reference cost after every iteration
            t_cost = max(abs(t[2] - curve[min(t[3], len(curve) - 1)]) for t in twins)
            window = _stop_window(ref_deltas, want_cost)
            slack = max(2 * window, 2 * t_shift)
            gap = _cost_gap(ref_deltas, n_iter, want_it, threshold=1e-6 * want_cost)
            data_shift = 0.0
            if dtype == "float32":            # the cost of the reference's end point on the data as the device holds them
                end = g["out_W_" + key]
                cost_at = lambda Xin: 0.5 * np.linalg.norm(Xin - end.dot(W.T)) ** 2 / Xin.shape[0]
                data_shift = abs(cost_at(Xn_r) - cost_at(Xn))
            tol = max(1e-10 * want_cost, 20 * t_cost, 20 * data_shift) + gap + _trace_noise(Xn, dtype)
            print("gpnh transform %s %s: n_iter %d (reference %d, window %d, the oracle's twins move by %d), "
                  "|dcost| %.2e (bound %.2e, of which iterations not run / run beyond %.2e)"
                  % (dtype, key, n_iter, int(want_it), window, t_shift, abs(cn - want_cost), tol, gap))
            assert abs(n_iter - int(want_it)) <= slack, (key, n_iter, want_it, window, t_shift)
            assert abs(cn - want_cost) <= tol, (key, abs(cn - want_cost), tol)
            step = np.abs(g["out_W_" + key] - g["out_Wprev_" + key]).max()
            shift = abs(n_iter - int(want_it))
            same_stop = [np.abs(t[0] - g["out_W_" + key]).max() for t in twins if t[3] == int(want_it)]
            wtol = max((shift + 1) * 2 * step, 1e-5, 20 * max(same_stop) if same_stop else 0.0)
            werr = np.abs(Wn - g["out_W_" + key]).max()
            print("    weights max |dW| %.2e (bound %.2e; the reference's last iteration moved them by %.1e)"
                  % (werr, wtol, step))
            assert werr <= wtol, (key, werr, wtol)
            _assert_simplex(Wn, 1e-12)
            assert np.abs(m.inverse_transform(Wn) - g["out_inverse_" + key]).max() <= 4 * wtol * np.abs(W).max()


# ---------------------------------------------------------------- bigger problems vs oracle
@pytest.mark.parametrize("dtype,k,rtol", [("float64", 32, 1e-9), ("float32", 32, 2e-5),
                                          ("float64", 40, 1e-9), ("float32", 5, 2e-5)])
def test_medium_problem_vs_oracle(cdr, orc, qp_kernel, dtype, k, rtol):
    """n = 3000, p = 700 (not a multiple of any tile): three production outer iterations
    from the same start; factors, costs, support and argmax against the oracle."""
    from convex_dim_red import archetypal_analysis as aa
    rng = np.random.RandomState(k)
    n, p = 3000, 700
    B = rng.standard_normal((k, p))
    Zt = orc.right_stochastic_matrix((n, k), rng) ** 4
    Zt /= Zt.sum(axis=1, keepdims=True)
    X = Zt.dot(B) + 0.05 * rng.standard_normal((n, p))
    if dtype == "float32":
        X = X.astype(np.float32)
    rs = np.random.RandomState(1)
    C0 = orc.right_stochastic_matrix((k, n), rs)
    Z0 = orc.right_stochastic_matrix((n, k), rs)
    kw = dict(tolerance=0, max_iterations=3, dictionary_solver_kwargs=dict(max_iterations=1),
              require_monotonic_cost_decrease=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Xd = X.astype(np.float64)
        wZ, wC, _, wcost, _, _, wdeltas = orc.iterate_aa(
            Xd, Z0.copy(), C0.copy(), np.ones(k), trace_XXt=(Xd * Xd).sum(), **kw)
        Z, C, _, cost, n_iter, _, deltas = aa._iterate_aa(
            X, Z0.copy(), C0.copy(), np.ones(k), dtype=dtype, **kw)
    assert abs(cost - wcost) < rtol * wcost
    assert np.abs(np.asarray(deltas) - np.asarray(wdeltas)).max() < 10 * rtol * wcost
    assert np.array_equal(C.argmax(axis=1), wC.argmax(axis=1))
    _assert_simplex(C, 1e-12)
    _assert_simplex(Z, 1e-12)
    if dtype == "float64":
        assert np.abs(C - wC).max() < 1e-7
        assert np.abs(Z - wZ).max() < 1e-4
        # support pattern (entries of rounding-dust size, ~1e-19, excluded on both sides)
        assert np.array_equal(C > 1e-15, wC > 1e-15)


def test_reconstruction_cost_matches_trace_form(cdr, orc):
    from convex_dim_red import _backend
    rng = np.random.RandomState(9)
    n, p, k = 1000, 300, 6
    X = rng.standard_normal((n, p)).astype(np.float32)
    C = orc.right_stochastic_matrix((k, n), rng)
    Z = orc.right_stochastic_matrix((n, k), rng)
    Xd = X.astype(np.float64)
    want = 0.5 * np.linalg.norm(Xd - Z.dot(C.dot(Xd))) ** 2 / n
    for dtype in ("float32", "float64"):
        with _backend.Context(dtype=dtype) as ctx:
            ctx.set_data(X)
            ctx.set_state(C, Z, np.ones(k))
            cost = ctx.prepare()
            rec = ctx.reconstruction_cost()
            tr = ctx.data_trace()
        assert abs(tr - (Xd * Xd).sum()) < 1e-10 * tr
        assert abs(rec - want) < (1e-6 if dtype == "float32" else 1e-12) * want
        assert abs(cost - want) < (1e-4 if dtype == "float32" else 1e-10) * want


# ---------------------------------------------------------------- building blocks
@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("k", [3, 32, 40, 64])
def test_gram_products_vs_numpy(cdr, orc, dtype, k):
    """Every GEMM / Gram kernel through aa_prepare: C X (reduce-over-rows), (CX)X' and
    X(X'Z) (row-local), Z'Z, (CX)(CX)', C XX'Z, and the cost assembled from them."""
    from convex_dim_red import _backend
    rng = np.random.RandomState(100 + k)
    n, p = 777, 333
    X = rng.standard_normal((n, p))
    if dtype == "float32":
        X = X.astype(np.float32)
    Xd = X.astype(np.float64)
    C = orc.right_stochastic_matrix((k, n), rng)
    Z = orc.right_stochastic_matrix((n, k), rng)
    alpha = rng.uniform(0.9, 1.1, size=k)
    tol = 1e-12 if dtype == "float64" else 3e-6
    with _backend.Context(dtype=dtype) as ctx:
        ctx.set_data(X)
        ctx.set_state(C, Z, alpha)
        cost = ctx.prepare()
        ZtZ, CKCt, CKZ, trace = ctx.grams()
        P = ctx.archetypes()
        C2, Z2, a2 = ctx.get_state()
    assert np.array_equal(C2, C) and np.array_equal(Z2, Z) and np.array_equal(a2, alpha)
    wP = C.dot(Xd)
    scale = np.abs(wP).max()
    assert np.abs(P - wP).max() < tol * scale * 10
    assert np.abs(ZtZ - Z.T.dot(Z)).max() < 1e-12 * n
    wCKCt = wP.dot(wP.T)
    assert np.abs(CKCt - wCKCt).max() < tol * np.abs(wCKCt).max() * 10
    wCKZ = C.dot(Xd.dot(Xd.T.dot(Z)))
    assert np.abs(CKZ - wCKZ).max() < tol * np.abs(wCKZ).max() * 10
    want = orc.kernel_aa_cost(Xd.dot(Xd.T), Z, C, alpha)
    assert abs(cost - want) < (1e-11 if dtype == "float64" else 1e-4) * want


@pytest.mark.parametrize("k", [2, 5, 17, 40, 64])
def test_qp_sizes_vs_oracle(cdr, orc, qp_kernel, k):
    from convex_dim_red import _backend
    rng = np.random.RandomState(k)
    n, p = 300, 2 * k + 5
    W = rng.standard_normal((k, p))
    Zt = orc.right_stochastic_matrix((n, k), rng) ** 3
    Zt /= Zt.sum(axis=1, keepdims=True)
    Xs = Zt.dot(W) + 0.05 * rng.standard_normal((n, p))
    A, B = W.dot(W.T), W.dot(Xs.T)
    Z0 = orc.right_stochastic_matrix((n, k), rng)
    for kw, tol in ((dict(max_iterations=1), 1e-11), (dict(max_iterations=4), 1e-9), ({}, 2e-6)):
        got, it = _backend.qp_batch(A, B, Z0, "kn", return_iters=True, **kw)
        want, wit = orc.qp_batch(A, B, Z0, "kn", return_iters=True, **kw)
        assert np.abs(got - want).max() < tol * max(1.0, np.abs(A).max()), (k, kw)
        _assert_simplex(got)
        if kw:
            assert np.array_equal(it, wit)


@pytest.mark.parametrize("k,n", [(6, 5), (13, 17), (31, 333), (32, 64)])
@pytest.mark.parametrize("memory", [3, 8, 12, 32])
def test_qp_nonmonotone_memory_vs_oracle(cdr, orc, qp_kernel, k, n, memory):
    """spg.py:310,341-344: the reference value of the Armijo test is the maximum of the last
    `memory` objective values.  Every mapping keeps that history in registers (8 entries; beyond
    that the wave-per-sample kernel, which holds 32, takes the update); fewer samples than a wave
    has slots, padded component counts and a full 32 are all in here."""
    from convex_dim_red import _backend
    rng = np.random.RandomState(100 * k + memory)
    p = 2 * k + 3
    W = rng.standard_normal((k, p)) * (1.0 + 5.0 * rng.rand(k, 1))      # uneven scales: back-tracking happens
    Zt = orc.right_stochastic_matrix((n, k), rng) ** 3
    Zt /= Zt.sum(axis=1, keepdims=True)
    Xs = Zt.dot(W) + 0.05 * rng.standard_normal((n, p))
    A, B = W.dot(W.T), W.dot(Xs.T)
    Z0 = orc.right_stochastic_matrix((n, k), rng)
    for kw, tol in ((dict(max_iterations=8), 1e-9), ({}, 2e-6)):
        got, it = _backend.qp_batch(A, B, Z0, "kn", return_iters=True, memory=memory, **kw)
        want, wit = orc.qp_batch(A, B, Z0, "kn", return_iters=True, memory=memory, **kw)
        assert np.abs(got - want).max() < tol * max(1.0, np.abs(A).max()), (k, kw)
        _assert_simplex(got)
        if kw:
            assert np.array_equal(it, wit)
        else:
            # runs of 100-500 passes on a badly scaled Hessian: pass counts of individual samples
            # move by tens with the last bit (the oracle's own 1-ulp twin does the same)
            assert abs(it.mean() - wit.mean()) < 0.1 * wit.mean() + 1


def test_qp_memory_beyond_the_register_budget_is_an_error(cdr, qp_kernel):
    """A reference-legal `memory` the kernels cannot hold is refused, never clamped."""
    from convex_dim_red import _backend
    rng = np.random.RandomState(3)
    A = np.eye(4); B = rng.standard_normal((4, 10)); Z0 = np.full((10, 4), 0.25)
    with pytest.raises(RuntimeError, match="memory"):
        _backend.qp_batch(A, B, Z0, "kn", memory=33)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("k", [40, 64])
def test_dictionary_update_wide_k_vs_oracle(cdr, orc, dtype, k):
    from convex_dim_red import _backend
    rng = np.random.RandomState(k)
    n, p = 600, 200
    X = rng.standard_normal((n, p))
    if dtype == "float32":
        X = X.astype(np.float32)
    Xd = X.astype(np.float64)
    C = orc.right_stochastic_matrix((k, n), rng)
    Z = orc.right_stochastic_matrix((n, k), rng)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = orc.update_aa_dictionary(Xd, C, np.ones(k), (Xd * Xd).sum(), Xd.dot(Xd.T.dot(Z)),
                                        Z.T.dot(Z), max_iterations=2)
    with _backend.Context(dtype=dtype) as ctx:
        ctx.set_data(X)
        ctx.set_state(C, Z, np.ones(k))
        ctx.prepare()
        st = ctx.dictionary_update(max_iterations=2)
        got = ctx.get_state()[0]
    tol = 1e-10 if dtype == "float64" else 1e-5
    assert abs(st.f - want[1]) < tol * abs(want[1])
    assert np.abs(got - want[0]).max() < tol
    assert (st.n_iter, st.n_feval) == (want[2], want[3])


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
@pytest.mark.parametrize("k", [7, 40])
def test_row_local_variants_agree(cdr, orc, variant, k):
    """The float32 row-local GEMM kernels (direct, wave-private LDS, and the block-tiled
    family with 32/64/128-column tiles, single or double buffered) give the same Gram
    products."""
    from convex_dim_red import _backend
    rng = np.random.RandomState(k)
    n, p = 1111, 450
    X = rng.standard_normal((n, p)).astype(np.float32)
    Xd = X.astype(np.float64)
    C = orc.right_stochastic_matrix((k, n), rng)
    Z = orc.right_stochastic_matrix((n, k), rng)
    _backend.set_option("row_local_variant", variant)
    try:
        with _backend.Context(dtype="float32") as ctx:
            ctx.set_data(X)
            ctx.set_state(C, Z, np.ones(k))
            ctx.prepare()
            ZtZ, CKCt, CKZ, trace = ctx.grams()
    finally:
        _backend.set_option("row_local_variant", -1)
    want = C.dot(Xd.dot(Xd.T.dot(Z)))
    assert np.abs(CKZ - want).max() < 3e-5 * np.abs(want).max()


@pytest.mark.parametrize("cap", [1, 3, 16, 1000])
def test_qp_pass_cap_invariance(cdr, orc, cap):
    """Handing samples from the lane-per-sample to the wave-per-sample QP kernel at any
    pass count does not change the result beyond rounding."""
    from convex_dim_red import _backend
    _backend.set_option("qp_mode", 2)
    rng = np.random.RandomState(11)
    n, k, p = 700, 12, 40
    W = rng.standard_normal((k, p))
    Zt = orc.right_stochastic_matrix((n, k), rng) ** 3
    Zt /= Zt.sum(axis=1, keepdims=True)
    Xs = Zt.dot(W) + 0.05 * rng.standard_normal((n, p))
    A, B = W.dot(W.T), W.dot(Xs.T)
    Z0 = orc.right_stochastic_matrix((n, k), rng)
    want, wit = orc.qp_batch(A, B, Z0, "kn", return_iters=True)
    _backend.set_option("qp_pass_cap", cap)
    try:
        got, it = _backend.qp_batch(A, B, Z0, "kn", return_iters=True)
    finally:
        _backend.set_option("qp_pass_cap", 24)
        _backend.set_option("qp_mode", 0)
    assert np.abs(got - want).max() < 2e-6
    assert abs(it.mean() - wit.mean()) < 0.05 * wit.mean()
    _assert_simplex(got)


def test_rccl_path_single_rank(cdr, orc):
    """The multi-rank code path (RCCL all-reduce at every splice point: split-row GEMM
    result, gathered per-rank reductions, gathered candidate lists of the projection, Grams,
    FurthestSum row broadcast) run on one GPU with a 1-rank communicator (AA_FORCE_RCCL=1):
    identical results to the direct path, for the list projection and for the iterative one;
    a list cap of 1 forces the overflow fallback (lists -> iterative passes)."""
    import os
    from convex_dim_red import _backend
    rng = np.random.RandomState(17)
    n, p, k = 900, 260, 6
    X = rng.standard_normal((n, p)).astype(np.float32)
    C = orc.right_stochastic_matrix((k, n), rng)
    Z = orc.right_stochastic_matrix((n, k), rng)

    def run(force, p2p=False, **opts):
        if force and not p2p:
            os.environ["AA_FORCE_RCCL"] = "1"
        for name, value in opts.items():
            _backend.set_option(name, value)
        try:
            with _backend.Context(dtype="float32") as ctx:
                if force and p2p:
                    ctx.p2p_init(0, 1, "test")          # the one-shot peer-to-peer transport, one rank
                elif force:
                    ctx.comm_init(_backend.comm_unique_id(), 0, 1)
                ctx.set_data(X)
                ctx.set_state(C, Z, np.ones(k))
                c0 = ctx.prepare()
                costs = ctx.outer_iterations(3, dict(max_iterations=1), {})
                # second call: the multi-rank path now knows its candidate lists are short and
                # defers the overflow check of each projection to the poll at the end of the call
                costs = np.concatenate([costs, ctx.outer_iterations(3, dict(max_iterations=1), {})])
                d = ctx.distance_column(5)
                Cf, Zf, _ = ctx.get_state()
                total = ctx.allreduce_host([1.5, 2.5])
            return c0, costs, d, Cf, Zf, total
        finally:
            os.environ.pop("AA_FORCE_RCCL", None)
            _backend.set_option("proj_mode", 0)
            _backend.set_option("proj_list_cap", 2048)
            _backend.set_option("proj_small", 1)
            _backend.set_option("proj_check", 0)
            _backend.set_option("pack_comm", 1)

    # bit-for-bit: both paths on the candidate-list projection (the single-rank default for
    # columns this short, the one-kernel threshold search, sums in another order)
    for opts in (dict(proj_small=0), dict(proj_mode=1)):
        a, b = run(False, **opts), run(True, **opts)
        assert a[0] == b[0] and np.array_equal(a[1], b[1])
        assert np.allclose(a[2], b[2], rtol=1e-6, atol=1e-6)   # the row travels through float64
        assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
        assert np.array_equal(b[5], [1.5, 2.5])
    # the same multi-rank code path over the peer-to-peer transport (round 4): same bits as over RCCL
    b2 = run(True, p2p=True, proj_small=0)
    b1 = run(True, proj_small=0)
    assert b1[0] == b2[0] and np.array_equal(b1[1], b2[1]) and np.array_equal(b1[3], b2[3]) and np.array_equal(b1[4], b2[4])
    assert np.array_equal(b2[5], [1.5, 2.5])
    # small reductions riding in the tail of the next all-reduce (pack_comm, round 4: 10 collectives per outer
    # iteration instead of 14) or travelling on their own: the same values reach the same consumers
    for p2p in (False, True):
        b3 = run(True, p2p=p2p, proj_small=0, pack_comm=0)
        assert b1[0] == b3[0] and np.array_equal(b1[1], b3[1]) and np.array_equal(b1[3], b3[3]) and np.array_equal(b1[4], b3[4])
    small = run(False)                                   # one-kernel threshold search: rounding level
    assert abs(small[0] - a[0]) < 1e-12 * abs(a[0]) and np.abs(small[1] - a[1]).max() < 1e-6 * abs(a[0])
    assert np.abs(small[3] - a[3]).max() < 1e-6 and np.abs(small[4] - a[4]).max() < 1e-5
    a, c = run(False, proj_small=0), run(True, proj_list_cap=1, proj_check=1)
    assert np.abs(np.asarray(a[1]) - np.asarray(c[1])).max() < 1e-9 * abs(a[0])
    assert np.abs(a[3] - c[3]).max() < 1e-12 and np.abs(a[4] - c[4]).max() < 1e-9
    # A slot shorter than the candidate lists but longer than the supports (round 4): the rank sends its
    # largest candidates and the largest one it left out; the threshold found on what was sent is
    # checked against that witness ON THE DEVICE, so the host neither looks (proj_check = 0: an
    # unconverged column would end the call with an error at the next poll) nor falls back.  Cold
    # projections collect every entry above max - 1 -- all 900 rows of a column here -- against
    # supports of a few dozen.  Same thresholds up to the summation order of the compacted lists.
    for cap in (128, 300):
        d = run(True, proj_list_cap=cap, proj_check=0)
        assert np.abs(np.asarray(a[1]) - np.asarray(d[1])).max() < 1e-12 * abs(a[0]), cap
        assert np.abs(a[3] - d[3]).max() < 1e-13 and np.abs(a[4] - d[4]).max() < 1e-9, cap


@pytest.mark.parametrize("n,k,dense", [(900, 6, False), (7000, 5, True), (7000, 40, True),
                                       (3000, 33, False)])
def test_projection_modes_agree(cdr, orc, n, k, dense):
    """Column simplex projection by candidate lists (one Newton step from the previous
    threshold, then the fixed point on the short list) and by iterative full passes give
    the same dictionary; `dense` starts from a dictionary whose columns have more than
    2048 non-zeros, so the list solver runs from global memory instead of LDS."""
    from convex_dim_red import _backend
    rng = np.random.RandomState(n + k)
    p = 120
    X = rng.standard_normal((n, p))
    C = orc.right_stochastic_matrix((k, n), rng)
    if not dense:
        C = C ** 6
        C /= C.sum(axis=1, keepdims=True)
    Z = orc.right_stochastic_matrix((n, k), rng)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = orc.update_aa_dictionary(X, C, np.ones(k), (X * X).sum(), X.dot(X.T.dot(Z)),
                                        Z.T.dot(Z), max_iterations=4)
    res = []
    for mode in (0, 1):
        _backend.set_option("proj_mode", mode)
        try:
            with _backend.Context(dtype="float64") as ctx:
                ctx.set_data(X)
                ctx.set_state(C, Z, np.ones(k))
                ctx.prepare()
                st = ctx.dictionary_update(max_iterations=4)
                res.append((ctx.get_state()[0], st.f, st.n_iter, st.n_feval, st.flags))
        finally:
            _backend.set_option("proj_mode", 0)
    for got in res:
        assert np.abs(got[0] - want[0]).max() < 1e-11
        assert abs(got[1] - want[1]) < 1e-10 * abs(want[1])
        assert got[2:4] == (want[2], want[3])
        assert np.abs(got[0].sum(axis=1) - 1).max() < 1e-13 and got[0].min() >= 0
    assert np.abs(res[0][0] - res[1][0]).max() < 1e-13
    assert np.array_equal(res[0][0] > 0, res[1][0] > 0) or \
        np.abs(res[0][0] - res[1][0])[(res[0][0] > 0) != (res[1][0] > 0)].max() < 1e-15


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-11), ("float32", 2e-5)])
def test_qp_tail_overlap_agrees(cdr, orc, dtype, tol):
    """Stragglers of the weights QP finishing on a side stream while Z'X is accumulated (their
    rows enter as a rank-m correction) give the same factors as the serial order.  A pass
    cap of 3 sends most samples down that path."""
    from convex_dim_red import _backend
    rng = np.random.RandomState(5)
    n, p, k = 2500, 300, 9
    X = rng.standard_normal((n, p))
    if dtype == "float32":
        X = X.astype(np.float32)
    C = orc.right_stochastic_matrix((k, n), rng)
    Z = orc.right_stochastic_matrix((n, k), rng)
    out = []
    _backend.set_option("qp_pass_cap", 3)
    _backend.set_option("qp_mode", 2)                  # the lane kernel, whatever the size
    try:
        for overlap in (0, 1):
            _backend.set_option("qp_overlap_tail", overlap)
            with _backend.Context(dtype=dtype) as ctx:
                ctx.set_data(X)
                ctx.set_state(C, Z, np.ones(k))
                ctx.prepare()
                costs = ctx.outer_iterations(4, dict(max_iterations=1), {})
                Cf, Zf, _ = ctx.get_state()
                out.append((np.asarray(costs), Cf, Zf, ctx.grams()))
    finally:
        _backend.set_option("qp_pass_cap", 24)
        _backend.set_option("qp_overlap_tail", 0)
        _backend.set_option("qp_mode", 0)
    a, b = out
    assert np.abs(a[0] - b[0]).max() <= tol * np.abs(a[0]).max()
    assert np.abs(a[1] - b[1]).max() <= tol and np.abs(a[2] - b[2]).max() <= 10 * tol
    for ga, gb in zip(a[3][:3], b[3][:3]):
        assert np.abs(ga - gb).max() <= tol * max(1.0, np.abs(ga).max())
    assert np.abs(b[2].sum(axis=1) - 1).max() < 1e-12 and b[2].min() >= 0


@pytest.mark.parametrize("n_outer", [8, 13])
def test_graph_replay_is_bit_identical(cdr, orc, n_outer):
    """aa_outer_iterations replays a captured pair of outer iterations (hipGraph); the same
    launches issued one by one give the same bits."""
    from convex_dim_red import _backend
    rng = np.random.RandomState(23)
    n, p, k = 1500, 200, 7
    X = rng.standard_normal((n, p)).astype(np.float32)
    C = orc.right_stochastic_matrix((k, n), rng)
    Z = orc.right_stochastic_matrix((n, k), rng)
    out = []
    try:
        for use in (0, 1):
            _backend.set_option("use_graph", use)
            with _backend.Context(dtype="float32") as ctx:
                ctx.set_data(X)
                ctx.set_state(C, Z, np.ones(k))
                ctx.prepare()
                costs = np.asarray(ctx.outer_iterations(n_outer, dict(max_iterations=1), {}))
                more = np.asarray(ctx.outer_iterations(2, dict(max_iterations=1), {}))
                Cf, Zf, _ = ctx.get_state()
                out.append((costs, more, Cf, Zf, ctx.cost()))
    finally:
        _backend.set_option("use_graph", 0)
    a, b = out
    assert a[0].size == 2 * n_outer
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert np.all(np.diff(np.asarray(a[0]).ravel()) < 1e-6)        # costs keep decreasing


def test_gemm_timing_counts_launches(cdr, orc):
    """aa_gemm_timing brackets every launch of the two pass kernels with HIP events: two of
    each per outer iteration once the dictionary products are warm; the timing mode does
    not change results."""
    from convex_dim_red import _backend
    rng = np.random.RandomState(2)
    n, p, k = 1200, 256, 5
    X = rng.standard_normal((n, p)).astype(np.float32)
    C = orc.right_stochastic_matrix((k, n), rng)
    Z = orc.right_stochastic_matrix((n, k), rng)
    outs = []
    for timed in (False, True):
        with _backend.Context(dtype="float32") as ctx:
            ctx.set_data(X)
            ctx.set_state(C, Z, np.ones(k))
            ctx.prepare()
            ctx.outer_iterations(1, dict(max_iterations=1), {})
            assert ctx.gemm_timing(timed) == (0.0, 0, 0.0, 0)
            costs = ctx.outer_iterations(3, dict(max_iterations=1), {})
            ms_r, n_r, ms_l, n_l = ctx.gemm_timing(False)
            outs.append(np.asarray(costs))
            if timed:
                assert (n_r, n_l) == (6, 6) and ms_r > 0 and ms_l > 0
            else:
                assert (n_r, n_l) == (0, 0)
    assert np.array_equal(outs[0], outs[1])


def test_qp_sample_order_does_not_change_results(cdr, orc):
    """The lane kernel takes the samples longest-first by their pass counts in the previous
    weights update; samples are independent, so the order cannot change a single bit."""
    from convex_dim_red import _backend
    rng = np.random.RandomState(31)
    n, p, k = 2300, 180, 11
    X = rng.standard_normal((n, p)).astype(np.float32)
    C = orc.right_stochastic_matrix((k, n), rng)
    Z = orc.right_stochastic_matrix((n, k), rng)
    out = []
    _backend.set_option("qp_mode", 2)
    try:
        for sort in (0, 1):
            _backend.set_option("qp_sort", sort)
            with _backend.Context(dtype="float32") as ctx:
                ctx.set_data(X)
                ctx.set_state(C, Z, np.ones(k))
                ctx.prepare()
                costs = np.asarray(ctx.outer_iterations(5, dict(max_iterations=1), {}))
                Cf, Zf, _ = ctx.get_state()
                out.append((costs, Cf, Zf))
    finally:
        _backend.set_option("qp_sort", 1)
        _backend.set_option("qp_mode", 0)
    for a, b in zip(*out):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("k", [7, 40])
@pytest.mark.parametrize("mode", [0, 2, 3])
def test_row_local_f64_variants_agree(cdr, orc, mode, k):
    """float64 data: the row-local GEMM on the f64 VALU (0), wave-streaming on the f64 matrix
    cores (2) and block-tiled on the matrix cores (3) give the same Gram products."""
    from convex_dim_red import _backend
    rng = np.random.RandomState(k)
    n, p = 1111, 450
    X = rng.standard_normal((n, p))
    C = orc.right_stochastic_matrix((k, n), rng)
    Z = orc.right_stochastic_matrix((n, k), rng)
    _backend.set_option("f64_mfma", mode)
    try:
        with _backend.Context(dtype="float64") as ctx:
            ctx.set_data(X)
            ctx.set_state(C, Z, np.ones(k))
            ctx.prepare()
            ZtZ, CKCt, CKZ, trace = ctx.grams()
    finally:
        _backend.set_option("f64_mfma", 1)
    want = C.dot(X.dot(X.T.dot(Z)))
    assert np.abs(CKZ - want).max() < 1e-12 * np.abs(want).max()
    assert np.abs(CKCt - C.dot(X).dot(C.dot(X).T)).max() < 1e-12 * np.abs(CKCt).max()


# ---------------------------------------------------------------- device-side loop control
@pytest.mark.parametrize("check_every", [1, 3, 8])
@pytest.mark.parametrize("criterion,tol", [("abs_delta_f", 1e-5), ("rel_delta_f", 1e-4)])
def test_device_loop_matches_host_loop(cdr, orc, check_every, criterion, tol):
    """aa_iterate (monotonicity check and stopping rule on the device, the host polling every
    `check_every` iterations, factors of the stopping iteration kept by a conditional snapshot)
    against the same loop driven from the host one update at a time
    (archetypal_analysis.py:586-663): same n_iter, same factors bit for bit, same costs."""
    from convex_dim_red import _backend
    rng = np.random.RandomState(41)
    n, p, k = 900, 64, 5
    B = rng.uniform(size=(k, p))
    X = orc.right_stochastic_matrix((n, k), rng).dot(B) + 0.01 * rng.standard_normal((n, p))
    C0 = orc.right_stochastic_matrix((k, n), rng)
    Z0 = orc.right_stochastic_matrix((n, k), rng)
    spg_kw, qp_kw = dict(max_iterations=1), {}
    done = {"abs_delta_f": lambda o, c: abs(c - o) < tol,
            "rel_delta_f": lambda o, c: abs((c - o) / max(abs(c), abs(o))) < tol}[criterion]
    with _backend.Context(dtype="float64") as ctx:
        ctx.set_data(X)
        ctx.set_state(C0, Z0, np.ones(k))
        cost = ctx.prepare()
        host_costs, n_host = [], -1
        for n_host in range(200):
            old = cost
            ctx.dictionary_update(**spg_kw)
            c1 = ctx.cost()
            ctx.weights_update(**qp_kw)
            cost = ctx.cost()
            host_costs += [c1, cost]
            if done(old, cost):
                break
        Ch, Zh, _ = ctx.get_state()
    assert 3 < n_host < 199
    with _backend.Context(dtype="float64") as ctx:
        ctx.set_data(X)
        ctx.set_state(C0, Z0, np.ones(k))
        cost0 = ctx.prepare()
        costs, st = ctx.iterate(cost0, 200, tol, criterion, True, True, True, spg_kw, qp_kw,
                                check_every=check_every)
        Cd, Zd, _ = ctx.get_state()
        after = ctx.cost()                                   # context consistent with the kept state
        more = ctx.outer_iterations(1, spg_kw, qp_kw)        # ... and usable for further work
    assert st.n_iter == n_host and st.converged == 1 and st.error_stage == 0
    assert st.reserved >= n_host + 1 and st.reserved <= n_host + check_every
    assert np.array_equal(Cd, Ch) and np.array_equal(Zd, Zh)
    assert np.array_equal(costs[1::2], host_costs[1::2])     # after-weights costs: same kernel
    assert np.abs(costs[0::2] - np.asarray(host_costs[0::2])).max() < 1e-12 * abs(cost0)
    assert abs(after - costs[-1]) < 1e-12 * abs(cost0) and st.cost == costs[-1]
    assert more[-1] <= after + 1e-12
    assert st.spg_flags & _backend.SPG_FLAG_MAX_ITER           # spg.py:278-281 on every update


def test_device_loop_reports_cost_increase(cdr, orc):
    """archetypal_analysis.py:167-174: a cost that goes up by more than the tolerance stops the
    loop with the stage that caused it.  Provoked with float32 data whose trace-form cost
    cancels tr(XX')/n = 5e6 down to ~1e2: its float32 noise (~0.5) is far above the tolerance,
    so some update soon appears to raise the cost."""
    from convex_dim_red import archetypal_analysis as aa
    rng = np.random.RandomState(3)
    n, p, k = 600, 512, 3
    X = (100.0 + rng.uniform(size=(n, p))).astype(np.float32)
    C0 = orc.right_stochastic_matrix((k, n), rng)
    Z0 = orc.right_stochastic_matrix((n, k), rng)
    kw = dict(tolerance=1e-9, max_iterations=80, dtype="float32",
              dictionary_solver_kwargs=dict(max_iterations=1))
    from convex_dim_red import _backend
    # the provocation needs the plain fp32 accumulation chain (one rounding per column of a sum
    # that grows to 5e6); with the 32-column pieces summed in float64 the noise is 100x smaller
    _backend.set_option("row_local_acc64", 0)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with pytest.raises(RuntimeError, match="factorization cost increased after (dictionary|weights) update"):
                aa._iterate_aa(X, Z0, C0, np.ones(k), **kw)
            # require_monotonic_cost_decrease=False: the same run goes through to the iteration cap
            out = aa._iterate_aa(X, Z0, C0, np.ones(k), require_monotonic_cost_decrease=False, **kw)
    finally:
        _backend.set_option("row_local_acc64", 1)
    assert out[4] == 79 and len(out[6]) == 80


def test_device_loop_update_switches(cdr, orc):
    """update_dictionary / update_weights = False (archetypal_analysis.py:534-541) leave that
    factor untouched and the loop still records two costs per iteration."""
    from convex_dim_red import archetypal_analysis as aa
    rng = np.random.RandomState(8)
    n, p, k = 400, 30, 4
    X = rng.uniform(size=(n, p))
    C0 = orc.right_stochastic_matrix((k, n), rng)
    Z0 = orc.right_stochastic_matrix((n, k), rng)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kw = dict(tolerance=0, max_iterations=4, dictionary_solver_kwargs=dict(max_iterations=1))
        Z, C, _, cost_w, n_iter, _, deltas = aa._iterate_aa(X, Z0, C0, np.ones(k),
                                                            update_dictionary=False, **kw)
        assert np.array_equal(C, C0) and not np.array_equal(Z, Z0) and n_iter == 3
        want = orc.iterate_aa(X, Z0.copy(), C0.copy(), np.ones(k), update_dictionary=False, **kw)
        assert abs(cost_w - want[3]) < 1e-9 * want[3]
        Z, C, _, cost_d, n_iter, _, deltas = aa._iterate_aa(X, Z0, C0, np.ones(k),
                                                            update_weights=False, **kw)
        assert np.array_equal(Z, Z0) and not np.array_equal(C, C0) and len(deltas) == 4
        want = orc.iterate_aa(X, Z0.copy(), C0.copy(), np.ones(k), update_weights=False, **kw)
        assert abs(cost_d - want[3]) < 1e-9 * want[3]


@pytest.mark.parametrize("lam", [0.0, 1.0])
def test_gpnh_device_loop_matches_host_loop(cdr, orc, lam):
    """aa_gpnh_iterate (Z'X, Cholesky solve of the k x k normal equations, X W, W'W, penalty, cost,
    QPs and loop control on the device) against the same loop driven from the host with
    numpy.linalg.lstsq (the reference's solver), and against the oracle."""
    from convex_dim_red import gpnh_convex_coding as gp
    rng = np.random.RandomState(12)
    n, p, k = 700, 45, 6
    W0 = rng.standard_normal((p, k))
    X = orc.right_stochastic_matrix((n, k), rng).dot(W0.T) + 0.1 * rng.standard_normal((n, p))
    Wi = 0.5 * rng.standard_normal((p, k))
    Zi = orc.right_stochastic_matrix((n, k), rng)
    # ten fixed iterations: rounding level (Cholesky against lstsq on a well conditioned system)
    kw = dict(lambda_W=lam, tolerance=0, max_iterations=10, stopping_criterion="rel_delta_f",
              weights_solver_kwargs=dict(max_iterations=1), require_monotonic_cost_decrease=False)
    want = orc.iterate_gpnh(X, Zi.copy(), Wi.copy(), **kw)
    outs = []
    for device in (True, False):
        gp._DEVICE_LOOP = device
        try:
            outs.append(gp._iterate_gpnh_convex_coding(X, Zi.copy(), Wi.copy(), **kw))
        finally:
            gp._DEVICE_LOOP = True
    for Z, W, cost, n_iter, _, deltas in outs:
        assert n_iter == 9 and len(deltas) == 10
        assert abs(cost - want[2]) < 1e-10 * want[2]
        assert np.abs(W - want[1]).max() < 1e-8 * np.abs(want[1]).max()
        assert np.abs(Z - want[0]).max() < 1e-8
        assert W.shape == (p, k)
        _assert_simplex(Z, 1e-12)
    assert np.abs(np.asarray(outs[0][5]) - np.asarray(outs[1][5])).max() < 1e-11
    # to the stopping rule: the iteration is not a contraction (see tests/test_gpu_configs.py:
    # ulp_perturbed), so the yardstick is the oracle's own sensitivity to a 1-ulp change of X
    kw = dict(lambda_W=lam, tolerance=1e-5, max_iterations=60, stopping_criterion="rel_delta_f",
              weights_solver_kwargs=dict(max_iterations=1))
    want = orc.iterate_gpnh(X, Zi.copy(), Wi.copy(), **kw)
    twin = orc.iterate_gpnh(X * (1.0 + 2e-16 * np.random.RandomState(5).standard_normal(X.shape)),
                            Zi.copy(), Wi.copy(), **kw)
    Z, W, cost, n_iter, _, deltas = gp._iterate_gpnh_convex_coding(X, Zi.copy(), Wi.copy(), **kw)
    assert abs(n_iter - want[3]) <= max(1, 2 * abs(twin[3] - want[3]))
    assert abs(cost - want[2]) < max(1e-9 * want[2], 20 * abs(twin[2] - want[2]), 2e-5 * want[2])
    assert len(deltas) == n_iter + 1
    _assert_simplex(Z, 1e-12)


def test_gpnh_unused_component_falls_back_to_lstsq(cdr, orc):
    """A weights matrix with an all-zero column makes Z'Z singular; with lambda_W = 0 the normal
    equations have no Cholesky factor, the device loop reports it and the host loop with
    numpy.linalg.lstsq (minimum-norm solution, the reference's behaviour) takes over."""
    from convex_dim_red import gpnh_convex_coding as gp
    rng = np.random.RandomState(2)
    n, p, k = 200, 12, 4
    X = rng.standard_normal((n, p))
    Zi = orc.right_stochastic_matrix((n, k), rng)
    Zi[:, 2] = 0.0
    Zi /= Zi.sum(axis=1, keepdims=True)
    Wi = rng.standard_normal((p, k))
    kw = dict(lambda_W=0.0, tolerance=0, max_iterations=1, update_weights=False,
              require_monotonic_cost_decrease=False)
    want = orc.iterate_gpnh(X, Zi.copy(), Wi.copy(), **kw)
    Z, W, cost, n_iter, _, deltas = gp._iterate_gpnh_convex_coding(X, Zi.copy(), Wi.copy(), **kw)
    assert np.array_equal(Z, Zi) and n_iter == 0
    assert abs(cost - want[2]) < 1e-10 * want[2]
    assert np.abs(W - want[1]).max() < 1e-9 * max(1.0, np.abs(want[1]).max())


@pytest.mark.parametrize("init", ["random", "furthest_sum"])
def test_kernel_aa_on_the_implicit_linear_kernel(cdr, orc, init):
    """KernelAA on K = X X' without forming K (fit_transform(X, features=True),
    aa_set_linear_kernel) against KernelAA on the explicit n x n matrix -- same RNG draws, same
    iterates, same n_iter -- and against the oracle's kernel-form loop from the same start."""
    rng = np.random.RandomState(11)
    n, p, k = 300, 40, 4
    B = rng.standard_normal((k, p))
    X = orc.right_stochastic_matrix((n, k), rng).dot(B) + 0.05 * rng.standard_normal((n, p))
    K = X.dot(X.T)
    kw = dict(n_components=k, init=init, tolerance=1e-9, max_iterations=40)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        explicit = cdr.KernelAA(random_state=np.random.RandomState(5), **kw)
        Ze = explicit.fit_transform(K)
        implicit = cdr.KernelAA(random_state=np.random.RandomState(5), **kw)
        Zi = implicit.fit_transform(X, features=True)
    assert implicit.n_iter == explicit.n_iter
    assert abs(implicit.cost - explicit.cost) < 1e-9 * explicit.cost
    assert np.abs(Zi - Ze).max() < 1e-6 and np.abs(implicit.dictionary - explicit.dictionary).max() < 1e-6
    assert np.array_equal(implicit.dictionary > 0, explicit.dictionary > 0)
    # five fixed iterations from one start: rounding level against the oracle's kernel form
    C0 = orc.right_stochastic_matrix((k, n), np.random.RandomState(2))
    Z0 = orc.right_stochastic_matrix((n, k), np.random.RandomState(3))
    # (fixed inner iteration counts too: QPs run to their 1e-6 stopping rule agree only to that)
    inner = dict(weights_solver_kwargs=dict(max_iterations=6), dictionary_solver_kwargs=dict(max_iterations=2))
    fixed = dict(tolerance=0, max_iterations=5, require_monotonic_cost_decrease=False, **inner)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = orc.iterate_kernel_aa(K, Z0.copy(), C0.copy(), np.ones(k), **fixed)
        est = cdr.KernelAA(n_components=k, init='custom', tolerance=0, max_iterations=5,
                           require_monotonic_cost_decrease=False, **inner)
        est.fit_transform(X, features=True, dictionary=C0.copy(), weights=Z0.copy(), alpha=np.ones(k))
    assert abs(est.cost - want[3]) < 1e-10 * want[3]
    assert np.abs(est.weights - want[0]).max() < 1e-8 and np.abs(est.dictionary - want[1]).max() < 1e-8


@pytest.mark.parametrize("form", ["data", "kernel"])
def test_scale_factors_on_device_match_host_spg(cdr, orc, form):
    """delta != 0 (archetypal_analysis.py:220-258,590-609): the k-vector scale-factor SPG inside the
    device loop (k_scale_factors_spg, one wave) against the same loop driven from the host with the
    generic host spg(), and against the oracle / the reference golden."""
    from convex_dim_red import archetypal_analysis as aa
    g = load_golden("iterate_aa")
    X, C0, Z0, a0 = g["in_X"], g["in_C0"], g["in_Z0"], g["in_alpha0"]
    # six fixed iterations: rounding level.  (Runs to the |delta cost| < 1e-6 rule stop on a flat
    # stretch of the cost curve -- 23 iterations in the reference, 26 with the last bits of alpha
    # rounded differently -- and alpha, which converges slowly, then differs by 1e-4: that tier is
    # covered by test_iterate_aa_traces_golden at the matching tolerance.)
    for skw, tol in ((dict(max_iterations=2), 1e-9), ({}, 2e-6)):
        # two SPG iterations per scale-factor update: the device kernel against the host spg() to
        # rounding carried through six outer iterations of simplex QPs (measured 3e-10 on the
        # kernel form, whose Gram has condition number ~1e6); default settings: each update runs to ||res|| < 1e-6, which is then the
        # accuracy two implementations share
        _check_scale_factors(aa, orc, X, Z0, C0, a0, form, skw, tol)


def _check_scale_factors(aa, orc, X, Z0, C0, a0, form, skw, tol):
    kw = dict(delta=0.1, tolerance=0, max_iterations=6, require_monotonic_cost_decrease=False,
              dictionary_solver_kwargs=dict(max_iterations=1), scale_factors_solver_kwargs=skw)
    outs = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for device in (True, False):
            aa._DEVICE_LOOP = device
            try:
                if form == "data":
                    outs.append(aa._iterate_aa(X, Z0.copy(), C0.copy(), a0.copy(), **kw))
                else:
                    outs.append(aa._iterate_kernel_aa(X.dot(X.T), Z0.copy(), C0.copy(), a0.copy(), **kw))
            finally:
                aa._DEVICE_LOOP = True
        if form == "data":
            want = orc.iterate_aa(X, Z0.copy(), C0.copy(), a0.copy(), **kw)
        else:
            want = orc.iterate_kernel_aa(X.dot(X.T), Z0.copy(), C0.copy(), a0.copy(), **kw)
    dev, host = outs
    assert dev[4] == host[4] == want[4] == 5                             # n_iter
    # device against host: the same state and kernels, only the k-vector SPG differs; against the
    # oracle the kernel form (Gram of condition number ~1e6 formed in NumPy) agrees to 2e-8
    assert abs(dev[3] - host[3]) < tol * dev[3] and abs(dev[3] - want[3]) < max(10 * tol, 1e-7) * dev[3]
    assert np.abs(dev[2] - host[2]).max() < 100 * tol and np.abs(dev[2] - want[2]).max() < 1000 * tol
    assert np.all(dev[2] >= 0.9 - 1e-15) and np.all(dev[2] <= 1.1 + 1e-15)
    assert np.abs(dev[2] - 1.0).max() > 1e-3                             # the scale factors did move
    assert np.abs(np.asarray(dev[6]) - np.asarray(host[6])).max() < tol * dev[3] * 10
    assert np.abs(dev[1] - want[1]).max() < 1e4 * tol and np.abs(dev[0] - want[0]).max() < 1e-4


# ---------------------------------------------------------------- FurthestSum on the device
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_furthest_sum_on_the_device(cdr, orc, dtype):
    """aa_furthest_sum (the whole selection as one chain of device launches, round 4) against the
    host-driven selection it replaces (one distance column per pick, the reference's list logic in
    convex_dim_red/furthest_sum.py) and against the reference's own selections on the golden
    dissimilarity matrices: same indices in the same order, with exclusions, 0 / 1 / 10 / 37 extra
    steps, k = 1 and k = 60; data with duplicated rows (shared maxima: the device raises its tie
    flag and the host's rule decides); the kernel form (explicit K)."""
    from convex_dim_red import _backend
    from convex_dim_red import archetypal_analysis as aa
    rng = np.random.RandomState(12)
    X = rng.standard_normal((700, 90))
    Xh = X.astype(np.float32) if dtype == "float32" else X

    def host(ctx, n, k, start, extra, exclude):
        aa._FURTHEST_SUM_ON_DEVICE = False
        try:
            return aa._furthest_sum_on_device(ctx, n, k, start, extra, exclude)
        finally:
            aa._FURTHEST_SUM_ON_DEVICE = True

    with _backend.Context(dtype=dtype) as ctx:
        ctx.set_data(Xh)
        for k, start, extra, exclude in ((4, 0, 1, []), (7, 59, 10, []), (1, 5, 3, []), (60, 3, 0, []),
                                         (5, 17, 37, [4, 99, 250, 699]), (3, 698, 10, [0])):
            want = host(ctx, 700, k, start, extra, np.asarray(exclude, dtype="i8"))
            got = ctx.furthest_sum(k, start, exclude, extra)
            if k == 1 and extra > 0:
                # the single point leaves, every candidate's running sum drops to exactly 0: a shared
                # maximum by construction, the device says so
                assert got is None
            else:
                assert got is not None and np.array_equal(got, want), (k, start, extra, exclude, got, want)
            assert np.array_equal(aa._furthest_sum_on_device(ctx, 700, k, start, extra, np.asarray(exclude, dtype="i8")), want)
    # duplicated rows: equal running sums -> the tie flag, and the dispatcher ends at the host's answer
    Xd = np.vstack([X[:50], X[:50], X[50:80]])
    Xdh = Xd.astype(np.float32) if dtype == "float32" else Xd
    with _backend.Context(dtype=dtype) as ctx:
        ctx.set_data(Xdh)
        want = host(ctx, 130, 6, 2, 10, np.array([], dtype="i8"))
        assert ctx.furthest_sum(6, 2, [], 10) is None
        assert np.array_equal(aa._furthest_sum_on_device(ctx, 130, 6, 2, 10, np.array([], dtype="i8")), want)
    # the kernel form: an explicit (n x n) kernel matrix on the device
    K = X[:300].dot(X[:300].T)
    Kh = K.astype(np.float32) if dtype == "float32" else K
    with _backend.Context(dtype=dtype) as ctx:
        ctx.set_data(Kh, form=_backend.FORM_KERNEL)
        for k, start, extra, exclude in ((4, 0, 1, []), (9, 299, 10, [7, 8])):
            want = host(ctx, 300, k, start, extra, np.asarray(exclude, dtype="i8"))
            got = ctx.furthest_sum(k, start, exclude, extra)
            assert got is not None and np.array_equal(got, want), (k, start, extra, exclude, got, want)


# ---------------------------------------------------------------- KernelAA on an implicit RBF kernel
def test_kernel_aa_on_the_implicit_rbf_kernel(cdr, orc):
    """SURVEY 8(f4), the alternative it names: KernelAA on K_ij = exp(-gamma ||x_i - x_j||^2) with K
    NEVER formed (aa_set_rbf_features: every C K / K Z is one fused distance + exp + multiply pass
    over the features).  Against (1) the ORACLE's kernel form on the explicit matrix
    (orc.iterate_kernel_aa, reference archetypal_analysis.py:399-531): fixed iterations from a
    custom start at rounding level; (2) the product's own explicit-kernel path on the same matrix:
    the estimators run to the stopping rule from FurthestSum and random starts -- same picks, same
    n_iter, costs and factors to rounding; (3) one K V product against NumPy."""
    from convex_dim_red import _backend
    rng = np.random.RandomState(21)
    n, p, k, gamma = 900, 12, 5, 0.35
    centers = rng.standard_normal((k, p)) * 2.0
    X = centers[rng.randint(k, size=n)] + 0.4 * rng.standard_normal((n, p))
    sq = (X * X).sum(axis=1)
    K = np.exp(-gamma * np.maximum(sq[:, None] + sq[None, :] - 2 * X.dot(X.T), 0.0))
    np.fill_diagonal(K, 1.0)
    C0 = orc.right_stochastic_matrix((k, n), rng)
    Z0 = orc.right_stochastic_matrix((n, k), rng)
    # (3) the product K Z through the kernel-form state of a context
    with _backend.Context(dtype="float64") as ctx:
        ctx.set_rbf_features(X, gamma)
        ctx.set_state(C0, Z0, np.ones(k))
        ctx.prepare()
        ZtZ, CKCt, CKZ, tr = ctx.grams()
        assert tr == float(n)
        assert np.abs(CKCt - C0.dot(K).dot(C0.T)).max() < 1e-12 * np.abs(CKCt).max()
        assert np.abs(CKZ - C0.dot(K).dot(Z0)).max() < 1e-12 * np.abs(CKZ).max()
        d = ctx.distance_column(17)
        assert np.abs(d - np.sqrt(np.maximum(2 - 2 * K[:, 17], 0))).max() < 1e-7 and d[17] == 0.0
    # (1) fixed iterations against the oracle on the explicit matrix
    kw = dict(tolerance=0, max_iterations=4, dictionary_solver_kwargs=dict(max_iterations=1),
              require_monotonic_cost_decrease=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wZ, wC, _, wcost, _, _, wdeltas = orc.iterate_kernel_aa(K, Z0.copy(), C0.copy(), np.ones(k), **kw)
        m = cdr.KernelAA(k, init="custom", tolerance=0, max_iterations=4, dictionary_solver_kwargs=dict(max_iterations=1),
                         require_monotonic_cost_decrease=False)
        W = m.fit_transform(X, features=True, kernel="rbf", gamma=gamma, dictionary=C0, weights=Z0, alpha=np.ones(k))
    assert abs(m.cost - wcost) < 1e-10 * abs(wcost)
    assert np.abs(np.asarray(m.cost_deltas) - np.asarray(wdeltas)).max() < 1e-9 * abs(wcost)
    assert np.abs(m.dictionary - wC).max() < 1e-8 and np.abs(W - wZ).max() < 1e-5
    _assert_simplex(W, 1e-12)
    _assert_simplex(m.dictionary, 1e-12)
    # (2) 25 iterations, implicit against the product's explicit-kernel path and the oracle, from the same
    # custom start; the yardstick of a run this long is the oracle's own response to a one-ulp change
    # of K (the explicit matrix and the fused exp differ in last bits)
    kw25 = dict(kw, max_iterations=25)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        base = orc.iterate_kernel_aa(K, Z0.copy(), C0.copy(), np.ones(k), **kw25)
        twins = oracle_twins(orc, lambda Kin: orc.iterate_kernel_aa(Kin, Z0.copy(), C0.copy(), np.ones(k), **kw25), K, "float64")
        t_cost = max(abs(t[3] - base[3]) for t in twins) / abs(base[3])
        t_C = max(np.abs(t[1] - base[1]).max() for t in twins)
        fits = []
        for data, extra in ((K, {}), (X, dict(features=True, kernel="rbf", gamma=gamma))):
            m = cdr.KernelAA(k, init="custom", tolerance=0, max_iterations=25, dictionary_solver_kwargs=dict(max_iterations=1),
                             require_monotonic_cost_decrease=False)
            m.fit_transform(data, dictionary=C0, weights=Z0, alpha=np.ones(k), **extra)
            fits.append(m)
    for m in fits:
        print("implicit RBF, 25 iterations: cost rel diff from the oracle %.2e (bound %.2e), dictionary %.2e (bound %.2e)"
              % (abs(m.cost - base[3]) / abs(base[3]), max(1e-10, 20 * t_cost), np.abs(m.dictionary - base[1]).max(),
                 max(1e-9, 20 * t_C)))
        assert abs(m.cost - base[3]) <= max(1e-10, 20 * t_cost) * abs(base[3])
        assert np.abs(m.dictionary - base[1]).max() <= max(1e-9, 20 * t_C)
        assert np.array_equal(m.dictionary.argmax(axis=1), base[1].argmax(axis=1))
    # FurthestSum on the implicit kernel.  Far-apart points of clustered data all sit at sqrt(2 - 2 exp(-big))
    # = sqrt(2) to the last bits, so the picks are decided by rounding and need not be the explicit
    # matrix's; they ARE the reference selection rule on the dissimilarities the context serves, and
    # those agree with the explicit ones to 1e-7.
    from convex_dim_red import archetypal_analysis as aa
    with _backend.Context(dtype="float64") as ci:
        ci.set_rbf_features(X, gamma)
        D = np.stack([ci.distance_column(j) for j in range(n)], axis=1)
        assert np.abs(D - np.sqrt(np.maximum(2 - 2 * K, 0))).max() < 1e-7
        for start, extra_steps in ((0, 10), (123, 3)):
            pi = aa._furthest_sum_on_device(ci, n, k, start, extra_steps, np.array([], dtype="i8"))
            assert np.array_equal(pi, cdr.furthest_sum(D, k, start, extra_steps=extra_steps)), (start, pi)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        bb = cdr.KernelAA(k, init="furthest_sum", random_state=3, tolerance=1e-5, max_iterations=40,
                          dictionary_solver_kwargs=dict(max_iterations=1))
        Wb = bb.fit_transform(X, features=True, kernel="rbf", gamma=gamma)
    assert bb.cost_deltas[0] < 0 and bb.cost > 0 and Wb.shape == (n, k)
    _assert_simplex(Wb, 1e-12)
    _assert_simplex(bb.dictionary, 1e-12)
    with pytest.raises(ValueError):
        cdr.KernelAA(k).fit_transform(K, kernel="rbf")          # rbf needs features=True
