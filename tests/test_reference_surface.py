"""The reference's own test properties (tests/test_archetypal_analysis.py,
tests/test_gpnh_convex_coding.py, tests/test_simplex_projection.py of the reference),
re-expressed against the MI355X package through the same private/public names the
reference tests import.  Tolerances are the reference's (1e-12 .. 1e-14: the float64
device path).  Needs a GPU."""
import warnings

import numpy as np
import pytest
from sklearn.utils import check_random_state

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def aa():
    from convex_dim_red import _backend, archetypal_analysis
    _backend.require_gpu()
    return archetypal_analysis


@pytest.fixture(scope="module")
def gp():
    from convex_dim_red import _backend, gpnh_convex_coding
    _backend.require_gpu()
    return gpnh_convex_coding


def _rsm(shape, rs):
    from convex_dim_red import right_stochastic_matrix
    return right_stochastic_matrix(shape, random_state=rs)


# ------------------------------------------------------------------ simplex projection
@pytest.mark.parametrize("dim", [1, 2, 5, 10, 100])
def test_projected_vectors_are_on_the_simplex(dim):
    from convex_dim_red.simplex_projection import simplex_project_vector
    rs = check_random_state(dim)
    for _ in range(5):
        x = rs.uniform(-3, 3, size=dim)
        y = simplex_project_vector(x)
        assert np.all(y >= 0) and abs(y.sum() - 1) < 1e-14
    inside = _rsm((1, dim), rs)[0]
    assert np.allclose(simplex_project_vector(inside), inside, rtol=0, atol=1e-15)


@pytest.mark.parametrize("shape", [(57, 5), (341, 317)])
def test_projected_rows_and_columns_sum_to_one(shape):
    from convex_dim_red import simplex_project_rows, simplex_project_columns
    rs = check_random_state(0)
    A = rs.uniform(-1, 1, size=shape)
    R = simplex_project_rows(A)
    assert np.all(R >= 0) and np.allclose(R.sum(axis=1), 1, 1e-14)
    Cc = simplex_project_columns(A)
    assert np.all(Cc >= 0) and np.allclose(Cc.sum(axis=0), 1, 1e-14)


# ------------------------------------------------------------------ kernel AA
def _kernel_problem(rs, n_features=10, n_components=5, n_samples=400, delta=0.0):
    X = rs.uniform(size=(n_samples, n_features))
    K = X.dot(X.T)
    C = _rsm((n_components, n_samples), rs)
    Z = _rsm((n_samples, n_components), rs)
    alpha = np.ones(n_components) if delta == 0 else rs.uniform(1 - delta, 1 + delta, n_components)
    return K, C, Z, alpha


@pytest.mark.parametrize("delta", [0.0, 0.1])
def test_single_dictionary_update_reduces_cost(aa, delta):
    rs = check_random_state(0)
    K, C, Z, alpha = _kernel_problem(rs, delta=delta)
    before = aa._kernel_aa_cost(K, Z, C, alpha)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        C1 = aa._update_kernel_aa_dictionary(K, C, alpha, np.trace(K), K.dot(Z), Z.T.dot(Z))
    assert aa._kernel_aa_cost(K, Z, C1, alpha) <= before
    assert np.allclose(C1.sum(axis=1), 1, 1e-12) and np.all(C1 >= 0)


@pytest.mark.parametrize("delta", [0.0, 0.1])
def test_single_weights_update_reduces_cost(aa, delta):
    rs = check_random_state(0)
    K, C, Z, alpha = _kernel_problem(rs, delta=delta)
    before = aa._kernel_aa_cost(K, Z, C, alpha)
    CK = C.dot(K)
    Z1 = aa._update_kernel_aa_weights(Z, alpha, CK, CK.dot(C.T))
    assert aa._kernel_aa_cost(K, Z1, C, alpha) <= before
    assert np.allclose(Z1.sum(axis=1), 1, 1e-12) and np.all(Z1 >= 0)


def _exact_problem(rs, n_features=30, n_components=10, n_samples=130):
    """Data that are exact convex combinations of `n_components` of the samples."""
    basis = rs.uniform(size=(n_components, n_features))
    Z = np.zeros((n_samples, n_components))
    Z[:n_components] = np.eye(n_components)
    Z[n_components:] = _rsm((n_samples - n_components, n_components), rs)
    X = Z.dot(basis)
    C = np.zeros((n_components, n_samples))
    C[np.arange(n_components), np.arange(n_components)] = 1
    assert np.linalg.norm(X - Z.dot(C.dot(X))) < 1e-12
    return X, X.dot(X.T), C, Z


def test_exact_solution_is_fixed_point_of_both_updates(aa):
    rs = check_random_state(0)
    X, K, C, Z = _exact_problem(rs)
    alpha = np.ones(C.shape[0])
    tolerance = 1e-12
    initial = aa._kernel_aa_cost(K, Z, C, alpha)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        C1 = aa._update_kernel_aa_dictionary(K, C, alpha, np.trace(K), K.dot(Z), Z.T.dot(Z))
    assert abs(aa._kernel_aa_cost(K, Z, C1, alpha) - initial) < tolerance
    assert np.allclose(C1, C, tolerance) and np.allclose(C1.sum(axis=1), 1, 1e-12)
    CK = C.dot(K)
    Z1 = aa._update_kernel_aa_weights(Z, alpha, CK, CK.dot(C.T))
    assert abs(aa._kernel_aa_cost(K, Z1, C, alpha) - initial) < tolerance
    assert np.allclose(Z1, Z, tolerance) and np.allclose(Z1.sum(axis=1), 1, 1e-12)


@pytest.mark.parametrize("which", ["dictionary", "weights"])
@pytest.mark.parametrize("delta", [0.0, 0.1])
def test_repeated_single_factor_updates_converge(aa, which, delta):
    rs = check_random_state(0)
    K, C, Z, alpha = _kernel_problem(rs, n_features=13, n_components=3, n_samples=100, delta=delta)
    before = aa._kernel_aa_cost(K, Z, C, alpha)
    max_iterations = 100
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Z1, C1, a1, cost, n_iter, _, deltas = aa._iterate_kernel_aa(
            K, Z.copy(), C.copy(), alpha.copy(), delta=delta,
            update_weights=(which == "weights"), update_dictionary=(which == "dictionary"),
            update_scale_factors=False, tolerance=1e-6, max_iterations=max_iterations)
    assert cost <= before and n_iter < max_iterations
    assert np.allclose(a1, alpha, 1e-12)
    if which == "dictionary":
        assert np.allclose(Z1, Z, 1e-12) and np.allclose(C1.sum(axis=1), 1, 1e-12)
    else:
        assert np.allclose(C1, C, 1e-12) and np.allclose(Z1.sum(axis=1), 1, 1e-12)


def test_finds_vertices_of_a_convex_hull():
    """KernelAA(init='custom') recovers the samples that span the hull (the reference's
    3- and 4-point hull tests, tests/test_archetypal_analysis.py:496-606)."""
    from convex_dim_red import KernelAA
    rs = check_random_state(0)
    basis = np.array([[0.0, 0.0], [1.0, 0.0], [1.0, 1.0], [0.0, 1.2]])
    k, n = 4, 123
    vertices = [8, 9, 56, 90]
    Zt = _rsm((n, k), rs) ** 2
    Zt /= Zt.sum(axis=1, keepdims=True)
    for i, v in enumerate(vertices):
        Zt[v] = 0
        Zt[v, i] = 1
    X = Zt.dot(basis)
    K = X.dot(X.T)
    C0 = _rsm((k, n), rs)
    Z0 = _rsm((n, k), rs)
    max_iter = 1000
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = KernelAA(n_components=k, delta=0, init="custom", tolerance=1e-10, max_iterations=max_iter)
        W = m.fit_transform(K, dictionary=C0, weights=Z0, alpha=np.ones(k))
    assert m.n_iter < max_iter
    assert np.allclose(m.dictionary.sum(axis=1), 1, 1e-12) and np.allclose(W.sum(axis=1), 1, 1e-12)
    assert sorted(m.dictionary.argmax(axis=1)) == vertices


# ------------------------------------------------------------------ GPNH
def test_gpnh_cost_is_zero_for_perfect_reconstruction(gp):
    rs = check_random_state(0)
    W = rs.uniform(size=(5, 3))
    Z = _rsm((30, 3), rs)
    assert abs(gp._gpnh_cost(Z.dot(W.T), Z, W, lambda_W=0)) < 1e-14


@pytest.mark.parametrize("lambda_W", [0.0, 3.2])
def test_gpnh_single_updates_reduce_cost(gp, lambda_W):
    rs = check_random_state(0)
    n, p, k = 100, 11, 5
    X = rs.uniform(size=(n, p))
    W = rs.uniform(size=(p, k))
    Z = _rsm((n, k), rs)
    before = gp._gpnh_cost(X, Z, W, lambda_W)
    GW = (4.0 / (p * k * (k - 1))) * (k * np.eye(k) - 1)
    W1 = gp._update_gpnh_dictionary(X, Z, Z.T.dot(Z), GW, lambda_W=lambda_W)
    assert gp._gpnh_cost(X, Z, W1, lambda_W) <= before
    Z1 = gp._update_gpnh_weights(X, Z, W)
    assert gp._gpnh_cost(X, Z1, W, lambda_W) <= before
    assert np.allclose(Z1.sum(axis=1), 1, 1e-14)


def test_gpnh_exact_solution_is_fixed_point(gp):
    rs = check_random_state(0)
    n, p, k = 50, 7, 3
    tolerance = 1e-6
    W = rs.uniform(size=(p, k))
    Z = _rsm((n, k), rs)
    X = Z.dot(W.T)
    initial = gp._gpnh_cost(X, Z, W, 0)
    GW = (4.0 / (p * k * (k - 1))) * (k * np.eye(k) - 1)
    W1 = gp._update_gpnh_dictionary(X, Z, Z.T.dot(Z), GW, lambda_W=0)
    assert np.allclose(W1, W, tolerance)
    assert abs(gp._gpnh_cost(X, Z, W1, 0) - initial) < tolerance
    Z1 = gp._update_gpnh_weights(X, Z, W)
    assert np.allclose(Z1, Z, tolerance)
    assert abs(gp._gpnh_cost(X, Z1, W, 0) - initial) < tolerance


@pytest.mark.parametrize("which", ["dictionary", "weights"])
@pytest.mark.parametrize("lambda_W", [0.0, 1.0])
def test_gpnh_repeated_updates_converge(gp, which, lambda_W):
    rs = check_random_state(0)
    n, p, k = 150, 9, 4
    X = rs.uniform(size=(n, p))
    W = rs.uniform(size=(p, k))
    Z = _rsm((n, k), rs)
    before = gp._gpnh_cost(X, Z, W, lambda_W)
    max_iterations = 100
    Z1, W1, cost, n_iter, _, deltas = gp._iterate_gpnh_convex_coding(
        X, Z, W, lambda_W=lambda_W, update_weights=(which == "weights"),
        update_dictionary=(which == "dictionary"), tolerance=1e-6, max_iterations=max_iterations)
    assert cost <= before and n_iter < max_iterations
    assert np.allclose(Z1.sum(axis=1), 1, 1e-12)
