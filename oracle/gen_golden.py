"""Generate tests/golden/*.npz from the REFERENCE itself.

TEST INFRASTRUCTURE.  Runs only in the build container (needs
/root/reference); the fixtures it writes are data (inputs + the reference's
outputs), committed under tests/golden/.  Usage:

    python oracle/gen_golden.py            # (re)writes tests/golden/*.npz
    python oracle/gen_golden.py qp gpnh    # only the named sections

Every array named ``in_*`` is an input, ``out_*`` an output of the reference
function named in the file's ``what`` entry (file:line citations there are
relative to /root/reference/src/convex_dim_red).
"""
import os
import sys
import warnings

import numpy as np
from numpy.random import RandomState

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.ref_loader import load_reference  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def save(name, what, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, what=np.array(what), **arrays)
    print("%-32s %7.1f KB" % (name, os.path.getsize(path) / 1024.0))


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = load_reference()
    aa = sys.modules["_ref_convex_dim_red.archetypal_analysis"]
    gp = sys.modules["_ref_convex_dim_red.gpnh_convex_coding"]
    rspg = sys.modules["_ref_convex_dim_red.spg"]
    rsp = sys.modules["_ref_convex_dim_red.simplex_projection"]
    warnings.simplefilter("ignore")
    only = sys.argv[1:]

    def aa_problem(seed, n, p, k, noise=0.01):
        rng = RandomState(seed)
        basis = rng.uniform(size=(k, p))
        Zt = ref.right_stochastic_matrix((n, k), random_state=rng)
        X = Zt.dot(basis) + noise * rng.randn(n, p)
        C0 = ref.right_stochastic_matrix((k, n), random_state=rng)
        Z0 = ref.right_stochastic_matrix((n, k), random_state=rng)
        return X, C0, Z0

    if not only or 'simplex' in only:
        # ---------------------------------------------------------------- simplex
        rng = RandomState(11)
        cases = {
            "exact_a": np.array([[0.5, 0.5], [0.5, 1.0], [0.0, -0.5]]),
            "exact_b": np.array([[0.8, 0.8], [0.0, 2.0], [0.5, -0.5]]),
            "single": np.array([[-0.5], [3.0], [1.0]]),
            "rand_57x5": rng.uniform(size=(57, 5)),
            "rand_101x317": rng.uniform(size=(101, 317)),
            "randn_64x32": 3.0 * rng.standard_normal((64, 32)),
            "ties": np.array([[0.25, 0.25, 0.25, 0.25, 0.25, 0.25],
                              [1.0, 1.0, 0.0, 0.0, -1.0, -1.0],
                              [2.0, 2.0, 2.0, -5.0, -5.0, 0.0],
                              [0.0, 0.0, 0.0, 0.0, 0.0, 0.0]]),
            "large": np.array([[1e8, 1e8 - 1.0, 3.0, -1e8], [1e-9, 2e-9, -1e-9, 0.0],
                               [1e3, 1e3 + 0.5, 1e3 - 0.5, 0.0]]),
            "in_simplex": np.array([[0.2, 0.3, 0.5], [1.0, 0.0, 0.0], [1 / 3., 1 / 3., 1 / 3.]]),
            "long_3x3000": rng.standard_normal((3, 3000)) * 0.01 + 1.0 / 3000,
            "long_sparse_3x4000": np.where(rng.uniform(size=(3, 4000)) < 0.002,
                                           rng.uniform(size=(3, 4000)), -rng.uniform(size=(3, 4000))),
        }
        arrays = {}
        for key, A in cases.items():
            arrays["in_" + key] = A
            arrays["out_" + key] = rsp.simplex_project_rows(A)
        save("simplex_rows", "simplex_project_rows simplex_projection.py:40-47", **arrays)

    if not only or 'qp' in only:
        # -------------------------------------------------------- quad_simplex_spg
        arrays = {}
        rng = RandomState(12)
        for k in (3, 8, 10, 32):
            n, p = 40, max(24, 2 * k)
            arch = rng.standard_normal((k, p))
            Zt = ref.right_stochastic_matrix((n, k), random_state=rng) ** 3
            Zt /= Zt.sum(axis=1, keepdims=True)
            Xs = Zt.dot(arch) + 0.05 * rng.standard_normal((n, p))
            A = arch.dot(arch.T)
            B = arch.dot(Xs.T)                       # k x n ; b_t = -B[:, t]
            Z0 = ref.right_stochastic_matrix((n, k), random_state=rng)
            for tag, kw in (("default", {}), ("one", dict(max_iterations=1)),
                            ("alpha0", dict(alpha0=0.5, max_iterations=5))):
                Z = np.stack([rspg.quad_simplex_spg(A, -B[:, t], Z0[t], **kw) for t in range(n)])
                arrays["out_Z_k%d_%s" % (k, tag)] = Z
            arrays["in_A_k%d" % k] = A
            arrays["in_B_k%d" % k] = B
            arrays["in_Z0_k%d" % k] = Z0
        save("quad_simplex_spg", "quad_simplex_spg spg.py:286-398; b_t = -B[:, t]", **arrays)

    if not only or 'furthest_sum' in only:
        # ------------------------------------------------------------ furthest_sum
        arrays = {}
        rng = RandomState(13)
        pts = rng.uniform(size=(60, 3))
        K = pts.dot(pts.T)
        d = np.diag(K)
        D = np.sqrt(np.tile(d, (60, 1)) - 2 * K + np.tile(d[:, None], (1, 60)))
        arrays["in_D_rand"] = D
        for k, start, extra in ((4, 0, 1), (4, 17, 10), (7, 59, 10), (1, 5, 3), (60, 3, 0)):
            arrays["out_rand_k%d_s%d_e%d" % (k, start, extra)] = np.asarray(
                ref.furthest_sum(D, k, start, None, extra))
        arrays["out_rand_excl"] = np.asarray(ref.furthest_sum(D, 5, 2, [0, 1, 7], 10))
        # ties: points on a regular grid -> many equal distance sums
        g = np.array([[i, j] for i in range(5) for j in range(5)], dtype=float)
        Dg = np.sqrt(((g[:, None, :] - g[None, :, :]) ** 2).sum(-1))
        arrays["in_D_grid"] = Dg
        for k, start, extra in ((4, 12, 0), (4, 12, 10), (6, 0, 10)):
            arrays["out_grid_k%d_s%d_e%d" % (k, start, extra)] = np.asarray(
                ref.furthest_sum(Dg, k, start, None, extra))
        save("furthest_sum", "furthest_sum furthest_sum.py:130-170", **arrays)

    if not only or 'dict_spg' in only:
        # ------------------------------------------- spg on the AA dictionary closure

        arrays = {}
        X, C0, Z0 = aa_problem(21, 120, 30, 4)
        alpha = np.ones(4)
        trX = np.trace(X.dot(X.T))
        XXtZ = X.dot(X.T.dot(Z0))
        ZtZ = Z0.T.dot(Z0)
        arrays.update(in_X=X, in_C0=C0, in_Z0=Z0)
        for tag, kw in (("one", dict(max_iterations=1)), ("five", dict(max_iterations=5)),
                        ("full", dict())):
            da = np.diag(alpha)
            XXtZD = XXtZ.dot(da)
            DZtZD = da.dot(ZtZ.dot(da))
            x, f, n_iter, n_feval = ref.spg(
                lambda c: aa._aa_dictionary_cost(X, c, trX, XXtZD, DZtZD),
                lambda c: aa._aa_dictionary_gradient(X, c, XXtZD, DZtZD),
                C0, project=ref.simplex_project_rows, **kw)
            arrays["out_C_" + tag] = x
            arrays["out_stats_" + tag] = np.array([f, n_iter, n_feval], dtype=np.float64)
        # kernel form
        K = X.dot(X.T)
        KZ = K.dot(Z0)
        for tag, kw in (("one", dict(max_iterations=1)), ("full", dict())):
            arrays["out_kernel_C_" + tag] = aa._update_kernel_aa_dictionary(
                K, C0, alpha, np.trace(K), KZ, ZtZ, **kw)
        # weights update (kernel form)
        CK = C0.dot(K)
        arrays["out_kernel_Z"] = aa._update_kernel_aa_weights(Z0, alpha, CK, CK.dot(C0.T))
        arrays["out_kernel_cost"] = np.array(aa._kernel_aa_cost(K, Z0, C0, alpha))
        save("aa_dictionary_spg", "spg on _aa_dictionary_cost/_gradient "
             "(archetypal_analysis.py:261-301,324-341) and kernel form (:273-321,369-396)", **arrays)

    if not only or 'iterate_aa' in only:
        # -------------------------------------------------- full _iterate_aa traces
        arrays = {}
        X, C0, Z0 = aa_problem(22, 150, 40, 5)
        arrays.update(in_X=X, in_C0=C0, in_Z0=Z0)
        for tag, kw in (("prod", dict(dictionary_solver_kwargs=dict(max_iterations=1))),
                        ("default", dict())):
            Z, C, al, cost, n_iter, _, deltas = aa._iterate_aa(
                X, Z0.copy(), C0.copy(), np.ones(5), delta=0, tolerance=1e-6,
                max_iterations=60, **kw)
            arrays["out_Z_" + tag] = Z
            arrays["out_C_" + tag] = C
            arrays["out_cost_" + tag] = np.array([cost, n_iter], dtype=np.float64)
            arrays["out_deltas_" + tag] = np.asarray(deltas)
        # first three outer steps separately (tight per-step comparison)
        Z, C = Z0.copy(), C0.copy()
        for step in range(3):
            Z, C, _, cost, _, _, _ = aa._iterate_aa(
                X, Z, C, np.ones(5), tolerance=0, max_iterations=1,
                dictionary_solver_kwargs=dict(max_iterations=1),
                require_monotonic_cost_decrease=False)
            arrays["out_step%d_Z" % step] = Z
            arrays["out_step%d_C" % step] = C
            arrays["out_step%d_cost" % step] = np.array(cost)
        # delta != 0 (scale factors)
        rs = RandomState(5)
        al0 = rs.uniform(0.9, 1.1, size=5)
        Z, C, al, cost, n_iter, _, deltas = aa._iterate_aa(
            X, Z0.copy(), C0.copy(), al0.copy(), delta=0.1, tolerance=1e-6, max_iterations=40,
            dictionary_solver_kwargs=dict(max_iterations=1))
        arrays.update(in_alpha0=al0, out_Z_delta=Z, out_C_delta=C, out_alpha_delta=al,
                      out_cost_delta=np.array([cost, n_iter], dtype=np.float64),
                      out_deltas_delta=np.asarray(deltas))
        save("iterate_aa", "_iterate_aa archetypal_analysis.py:534-670 from custom starts", **arrays)

    if not only or 'iterate_kernel_aa' in only:
        # ------------------------------------------- full _iterate_kernel_aa traces
        arrays = {}
        X, C0, Z0 = aa_problem(23, 100, 12, 4)
        K = X.dot(X.T)
        arrays.update(in_K=K, in_C0=C0, in_Z0=Z0)
        for tag, kw in (("prod", dict(dictionary_solver_kwargs=dict(max_iterations=1))),
                        ("default", dict())):
            Z, C, al, cost, n_iter, _, deltas = aa._iterate_kernel_aa(
                K, Z0.copy(), C0.copy(), np.ones(4), delta=0, tolerance=1e-6,
                max_iterations=40, **kw)
            arrays["out_Z_" + tag] = Z
            arrays["out_C_" + tag] = C
            arrays["out_cost_" + tag] = np.array([cost, n_iter], dtype=np.float64)
            arrays["out_deltas_" + tag] = np.asarray(deltas)
        save("iterate_kernel_aa", "_iterate_kernel_aa archetypal_analysis.py:399-531", **arrays)

    if not only or 'aa_estimator' in only:
        # ---------------------------------------------- estimator-level known answers
        arrays = {}
        rng = RandomState(0)
        basis = rng.uniform(size=(3, 50))
        Zt = ref.right_stochastic_matrix((200, 3), random_state=rng)
        X = Zt.dot(basis) + 0.01 * rng.randn(200, 50)
        arrays["in_X"] = X
        for init in ("furthest_sum", "random"):
            for tag, dkw in (("one", dict(max_iterations=1)), ("full", dict())):
                m = ref.ArchetypalAnalysis(3, init=init, random_state=0, tolerance=1e-6,
                                           max_iterations=1000, dictionary_solver_kwargs=dkw)
                W = m.fit_transform(X)
                key = "%s_%s" % (init, tag)
                arrays["out_cost_" + key] = np.array([m.cost, m.n_iter], dtype=np.float64)
                arrays["out_argmax_" + key] = m.dictionary.argmax(axis=1)
                arrays["out_W_" + key] = W
                arrays["out_archetypes_" + key] = m.archetypes
                if key == "furthest_sum_one":
                    Xn = X[:25] + 0.0
                    Wn, cn = m.transform(Xn)
                    arrays["out_transform_W"] = Wn
                    arrays["out_transform_cost"] = np.array(cn)
                    arrays["out_inverse"] = m.inverse_transform(Wn)
        save("aa_estimator", "ArchetypalAnalysis.fit_transform/transform "
             "archetypal_analysis.py:1026-1215; SURVEY 8c recipe", **arrays)

        # KernelAA hull tests (tests/test_archetypal_analysis.py:496-606 recipe class)
        arrays = {}
        rng = RandomState(31)
        k, n, p = 3, 50, 2
        basis = np.array([[0.0, 0.0], [1.0, 0.0], [0.3, 1.0]])
        Zt = ref.right_stochastic_matrix((n, k), random_state=rng)
        hull_idx = np.array([5, 27, 32])
        for i, h in enumerate(hull_idx):
            Zt[h] = 0
            Zt[h, i] = 1
        X = Zt.dot(basis)
        K = X.dot(X.T)
        C0 = ref.right_stochastic_matrix((k, n), random_state=rng)
        Z0 = ref.right_stochastic_matrix((n, k), random_state=rng)
        m = ref.KernelAA(k, delta=0, init="custom", tolerance=1e-10, max_iterations=500)
        W = m.fit_transform(K, dictionary=C0, weights=Z0, alpha=np.ones(k))
        arrays.update(in_K=K, in_C0=C0, in_Z0=Z0, out_W=W, out_C=m.dictionary,
                      out_argmax=m.dictionary.argmax(axis=1),
                      out_cost=np.array([m.cost, m.n_iter], dtype=np.float64))
        save("kernel_aa_estimator", "KernelAA.fit_transform archetypal_analysis.py:773-895, "
             "init='custom', archetypes at samples 5, 27, 32", **arrays)

    if not only or 'gpnh' in only:
        # ------------------------------------------------------------------- GPNH
        arrays = {}
        rng = RandomState(41)
        n, p, k = 120, 17, 4
        W0 = rng.standard_normal((p, k))
        Zt = ref.right_stochastic_matrix((n, k), random_state=rng)
        X = Zt.dot(W0.T) + 0.1 * rng.randn(n, p)
        Wi = rng.standard_normal((p, k))
        Zi = ref.right_stochastic_matrix((n, k), random_state=rng)
        arrays.update(in_X=X, in_W0=Wi, in_Z0=Zi)
        for lam in (0.0, 1.0):
            tag = "lam%d" % int(lam)
            arrays["out_cost0_" + tag] = np.array(gp._gpnh_cost(X, Zi, Wi, lam))
            GW = (4.0 / (p * k * (k - 1))) * (k * np.eye(k) - 1)
            arrays["out_Wupd_" + tag] = gp._update_gpnh_dictionary(X, Zi, Zi.T.dot(Zi), GW, lam)
            for wtag, wkw in (("one", dict(max_iterations=1)), ("full", dict())):
                Z, W, cost, n_iter, _, deltas = gp._iterate_gpnh_convex_coding(
                    X, Zi.copy(), Wi.copy(), lambda_W=lam, tolerance=1e-6, max_iterations=200,
                    stopping_criterion="rel_delta_f", weights_solver_kwargs=wkw)
                key = "%s_%s" % (tag, wtag)
                arrays["out_Z_" + key] = Z
                arrays["out_W_" + key] = np.ascontiguousarray(W)
                arrays["out_cost_" + key] = np.array([cost, n_iter], dtype=np.float64)
                arrays["out_deltas_" + key] = np.asarray(deltas)
        arrays["out_Zupd"] = gp._update_gpnh_weights(X, Zi, Wi)
        save("gpnh", "gpnh_convex_coding.py:199-402 from custom starts", **arrays)

        # GPNH estimator known answers (SURVEY 8c recipe, shortened to n=300)
        arrays = {}
        rng = RandomState(0)
        W0 = rng.standard_normal((30, 5))
        Zt = ref.right_stochastic_matrix((300, 5), random_state=rng)
        X = Zt.dot(W0.T) + 0.1 * rng.randn(300, 30)
        arrays["in_X"] = X
        for lam in (0.0, 1.0):
            for init in ("random", "furthest_sum"):
                m = ref.GPNHConvexCoding(5, lambda_W=lam, init=init, tolerance=1e-6,
                                         max_iterations=3000, stopping_criterion="rel_delta_f",
                                         random_state=0,
                                         weights_solver_kwargs=dict(max_iterations=1))
                Wt = m.fit_transform(X)
                key = "lam%d_%s" % (int(lam), init)
                arrays["out_cost_" + key] = np.array([m.cost, m.n_iter], dtype=np.float64)
                arrays["out_W_" + key] = np.ascontiguousarray(m.dictionary)
                arrays["out_Z_" + key] = Wt
                arrays["out_deltas_" + key] = np.asarray(m.cost_deltas)
        save("gpnh_estimator", "GPNHConvexCoding.fit_transform gpnh_convex_coding.py:501-606", **arrays)

    if not only or 'gpnh_transform' in only:
        # GPNHConvexCoding.transform (gpnh_convex_coding.py:623-652): weights of new samples for the
        # fitted dictionary -- fresh random weights from the estimator's generator (its state
        # after the fit), then the weights-only loop to the stopping rule
        arrays = {}
        rng = RandomState(0)
        W0 = rng.standard_normal((30, 5))
        Zt = ref.right_stochastic_matrix((300, 5), random_state=rng)
        X = Zt.dot(W0.T) + 0.1 * rng.randn(300, 30)
        Xn = Zt[:40][::-1].dot(W0.T) + 0.1 * rng.randn(40, 30)
        arrays.update(in_X=X, in_Xnew=Xn)
        for lam in (0.0, 1.0):
            for wtag, wkw in (("one", dict(max_iterations=1)), ("full", dict())):
                m = ref.GPNHConvexCoding(5, lambda_W=lam, init="random", tolerance=1e-6,
                                         max_iterations=400, stopping_criterion="rel_delta_f",
                                         random_state=0, weights_solver_kwargs=wkw)
                m.fit_transform(X)
                key = "lam%d_%s" % (int(lam), wtag)
                # the transform's loop from its own start, stage by stage: the start weights the
                # estimator's generator gives next (drawn from a copy of its state), the weights
                # after 1 and 3 weights-only iterations, and the cost changes of the run to the
                # stopping rule (transform() itself keeps none of these)
                import copy
                rs2 = copy.deepcopy(m.random_state)
                Z0n = gp._initialize_gpnh_convex_coding_weights(Xn, 5, init="random", random_state=rs2)
                arrays["out_start_" + key] = Z0n
                for iters in (1, 3):
                    Zi = gp._iterate_gpnh_convex_coding(
                        Xn, Z0n.copy(), m.dictionary.copy(), lambda_W=lam, update_dictionary=False,
                        update_weights=True, tolerance=0, max_iterations=iters,
                        stopping_criterion="rel_delta_f", weights_solver_kwargs=wkw)[0]
                    arrays["out_W%d_%s" % (iters, key)] = Zi
                tr = gp._iterate_gpnh_convex_coding(
                    Xn, Z0n.copy(), m.dictionary.copy(), lambda_W=lam, update_dictionary=False,
                    update_weights=True, tolerance=1e-6, max_iterations=400,
                    stopping_criterion="rel_delta_f", weights_solver_kwargs=wkw)
                arrays["out_trace_" + key] = np.array([tr[2], tr[3]], dtype=np.float64)   # cost, n_iter
                arrays["out_deltas_" + key] = np.asarray(tr[5])
                if tr[3] >= 1:      # the weights one iteration before the stop: how far an iteration still moves them
                    arrays["out_Wprev_" + key] = gp._iterate_gpnh_convex_coding(
                        Xn, Z0n.copy(), m.dictionary.copy(), lambda_W=lam, update_dictionary=False,
                        update_weights=True, tolerance=0, max_iterations=int(tr[3]),
                        stopping_criterion="rel_delta_f", weights_solver_kwargs=wkw)[0]
                Wn, cn = m.transform(Xn)
                assert np.array_equal(Wn, tr[0]) and cn == tr[2]     # the same loop from the same start
                arrays["out_fit_" + key] = np.array([m.cost, m.n_iter], dtype=np.float64)
                arrays["out_dictionary_" + key] = np.ascontiguousarray(m.dictionary)   # the transform's input
                arrays["out_W_" + key] = Wn
                arrays["out_cost_" + key] = np.array([cn, m.n_iter], dtype=np.float64)
                arrays["out_inverse_" + key] = m.inverse_transform(Wn)
        save("gpnh_transform", "GPNHConvexCoding.transform / inverse_transform "
             "gpnh_convex_coding.py:623-668 after fit_transform", **arrays)


if __name__ == "__main__":
    main()
