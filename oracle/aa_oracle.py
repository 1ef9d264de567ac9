"""CPU oracle for the archetypal-analysis / GPNH convex-coding solver path.

TEST INFRASTRUCTURE -- NOT THE PRODUCT.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product (``matrix-factorization-case-studies_amd/convex_dim_red``)
never does, and fails loudly when its HIP library is missing.

Parity status: PINNED -- ``tests/test_oracle_golden.py`` checks every function
here against outputs of the reference itself (``tests/golden/*.npz``, generated
by ``oracle/gen_golden.py`` from ``/root/reference/src/convex_dim_red``).

A float64 NumPy restatement of the reference's exact operation sequence
(citations relative to ``/root/reference/src/convex_dim_red``).  The two serial
inner loops of the reference (sort-based simplex projection and the per-sample
simplex QP; numba-compiled there) are restated twice: in pure NumPy/Python here
(``*_py``; small cases) and in plain C (``oracle/c/aa_oracle.c``; loaded through
ctypes when built) for sizes where the Python loop would take minutes.
"""
import ctypes
import os
import time
import warnings

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# AA_ORACLE_CLIB: another build of oracle/c/aa_oracle.c (the AddressSanitizer build of
# tests/test_oracle_golden.py::test_oracle_c_under_address_sanitizer)
_CLIB_PATH = os.environ.get("AA_ORACLE_CLIB") or os.path.join(_HERE, "_build", "libaa_oracle.so")


# --------------------------------------------------------------------------
# solver parameter handling (defaults: spg.py:287-291, archetypal_analysis.py:372-383)
# --------------------------------------------------------------------------
QP_DEFAULTS = dict(gamma=1e-4, memory=1, sigma_one=0.1, sigma_two=0.9,
                   lambda_min=1e-10, alpha0=-1.0, alpha_min=1e-5, alpha_max=1e3,
                   epsilon_one=1e-10, epsilon_two=1e-6,
                   max_iterations=1000, max_feval=2000)


def qp_params(**kw):
    p = dict(QP_DEFAULTS)
    for key, val in kw.items():
        if key in p:
            p[key] = val
    return p


class _CParams(ctypes.Structure):
    _fields_ = [("gamma", ctypes.c_double), ("memory", ctypes.c_int),
                ("sigma_one", ctypes.c_double), ("sigma_two", ctypes.c_double),
                ("lambda_min", ctypes.c_double), ("alpha0", ctypes.c_double),
                ("alpha_min", ctypes.c_double), ("alpha_max", ctypes.c_double),
                ("epsilon_one", ctypes.c_double), ("epsilon_two", ctypes.c_double),
                ("max_iterations", ctypes.c_int), ("max_feval", ctypes.c_int)]


# --------------------------------------------------------------------------
# yardstick mode for the float32 legs of the GPU tests: what rounding the non-data operand
# of every big contraction against X to float32, and accumulating the reduce-over-rows pass (_rr)
# and the 32-column pieces of the row-local pass (_rl) in float32, does to an otherwise exact
# (float64) run.
# The HIP path's float32 mode feeds its four passes over X with float32 operands (the dictionary /
# weights / search direction as MFMA A-operands of the reduce-over-rows pass, C X and X'Z as
# operands of the row-local pass); an archetype row C X is an average over thousands of samples, so
# rounding IT moves it far more than float32-sized noise on the samples does.  Off (None) the
# functions below are the reference's op sequence, bit for bit.
# --------------------------------------------------------------------------
_OPERAND_DTYPE = None


class operand_rounding(object):
    """``with operand_rounding(np.float32): ...`` -- see above.  Test yardstick only."""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global _OPERAND_DTYPE
        self.saved, _OPERAND_DTYPE = _OPERAND_DTYPE, self.dtype
        _rr.calls = 0                              # the noise of a run is reproducible
        return self

    def __exit__(self, *exc):
        global _OPERAND_DTYPE
        _OPERAND_DTYPE = self.saved
        return False


def _op(a):
    if _OPERAND_DTYPE is None:
        return a
    return np.asarray(a).astype(_OPERAND_DTYPE).astype(np.float64)


def _rr(A, X):
    """A . X for a k x n matrix A against the data (the reduce-over-rows pass).  Yardstick mode: A is
    rounded to the operand dtype and the result carries the error of a float32 accumulation chain,
    modelled as Gaussian noise of 1e-7 |A| . |X| per entry -- the middle of the 5e-8 ... 1.9e-7 the
    float32 pass kernel measures against float64 NumPy (tests/test_gpu_longrun.py::
    test_pass_kernels_against_numpy pins it below 2e-7).  With c >= 0 and data of mixed signs
    |A| . |X| is several times |A . X|, so this is the largest float32 effect on C X."""
    if _OPERAND_DTYPE is None:
        return A.dot(X)
    Ar = _op(A)
    out = Ar.dot(X)
    if np.dtype(_OPERAND_DTYPE) == np.float32:
        _rr.calls += 1
        noise = np.random.RandomState(_rr.calls).standard_normal(out.shape)
        out = out + 1e-7 * np.abs(Ar).dot(np.abs(X)) * noise
    return out


_rr.calls = 0


def _rl(X, B):
    """X . B for the data against a p x k matrix B (the row-local pass).  Yardstick mode: B is rounded
    to the operand dtype and the result carries the error of the float32 accumulation chain inside
    every 32-column piece (the pieces themselves are summed in float64):
    independent Gaussian noise of 4e-8 x (sum of |x||b| over the piece) per piece -- at p = 4096
    (128 pieces) an rms of 3.5e-9 of sum |x||b|, which is what k_row_local_f32_dma measures against
    float64 NumPy; at p <= 32 the whole of 4e-8 (one chain)."""
    if _OPERAND_DTYPE is None:
        return X.dot(B)
    Br = _op(B)
    out = X.dot(Br)
    if np.dtype(_OPERAND_DTYPE) == np.float32:
        _rr.calls += 1
        var = np.zeros_like(out)
        aX, aB = np.abs(X), np.abs(Br)
        for c0 in range(0, X.shape[1], 32):
            var += aX[:, c0:c0 + 32].dot(aB[c0:c0 + 32]) ** 2
        out = out + 4e-8 * np.sqrt(var) * np.random.RandomState(_rr.calls).standard_normal(out.shape)
    return out


_clib = None


def clib():
    """The C restatement (oracle/c/aa_oracle.c), or None when not built."""
    global _clib
    if _clib is None and os.path.exists(_CLIB_PATH):
        lib = ctypes.CDLL(_CLIB_PATH)
        dp = ctypes.POINTER(ctypes.c_double)
        lib.orc_simplex_project_rows.argtypes = [dp, dp, ctypes.c_long, ctypes.c_long]
        lib.orc_simplex_project_rows.restype = None
        lib.orc_qp_batch.argtypes = [dp, dp, ctypes.c_long, ctypes.c_long, dp, dp,
                                     ctypes.c_long, ctypes.c_int,
                                     ctypes.POINTER(_CParams), ctypes.POINTER(ctypes.c_int)]
        lib.orc_qp_batch.restype = None
        _clib = lib
    return _clib


def _dptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


# --------------------------------------------------------------------------
# simplex projection (simplex_projection.py:13-47)
# --------------------------------------------------------------------------
def simplex_project_vector_py(x):
    """simplex_projection.py:13-27, same arithmetic (including the O(m^2)
    re-summation of the m largest entries at :23, so results are bit-identical
    to the reference)."""
    x = np.asarray(x, dtype=np.float64)
    s = np.sort(x)
    n = s.size
    t_hat = 0
    for i in range(n - 2, -2, -1):
        m = n - 1 - i
        t_hat = (s[-m:].sum() - 1) / m
        if t_hat >= s[i]:
            break
    return np.fmax(x - t_hat, 0)


def simplex_project_rows_py(A):
    A = np.asarray(A, dtype=np.float64)
    return np.stack([simplex_project_vector_py(r) for r in A]) if A.shape[0] else A.copy()


def simplex_project_rows_vec(A):
    """Vectorised restatement (cumulative sums of the descending sort): the same
    threshold rule, summation order differs from :23 by rounding only."""
    A = np.asarray(A, dtype=np.float64)
    rows, n = A.shape
    s = np.sort(A, axis=1)[:, ::-1]                     # descending
    cs = np.cumsum(s, axis=1)
    m = np.arange(1, n + 1, dtype=np.float64)
    t = (cs - 1.0) / m                                  # t for support size m
    nxt = np.concatenate([s[:, 1:], s[:, :1]], axis=1)  # sorted[i]; wraps to largest at m = n
    stop = t >= nxt
    stop[:, -1] = True
    first = stop.argmax(axis=1)
    t_hat = t[np.arange(rows), first]
    return np.fmax(A - t_hat[:, None], 0)


def simplex_project_rows(A):
    """Dispatcher used by the L1 loops: C when built, vectorised NumPy otherwise."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    lib = clib()
    if lib is None:
        return simplex_project_rows_vec(A)
    out = np.empty_like(A)
    lib.orc_simplex_project_rows(_dptr(A), _dptr(out), A.shape[0], A.shape[1])
    return out


def simplex_project_columns(A):
    return simplex_project_rows(np.asarray(A).T).T.copy()


# --------------------------------------------------------------------------
# SPG pieces (spg.py)
# --------------------------------------------------------------------------
def line_search_step_length(lam, delta, f_old, f_new, sigma_one=0.1, sigma_two=0.9):
    """spg.py:19-33 (sigma_one is an ABSOLUTE lower bound, :28)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        trial = -0.5 * lam ** 2 * delta / (f_new - f_old - lam * delta)
    if sigma_one <= trial <= sigma_two * lam:
        return trial
    return 0.5 * lam


def cauchy_step_size(beta, sksk, alpha_min=1e-3, alpha_max=1e3):
    """spg.py:36-43."""
    if beta <= 0:
        return alpha_max
    return min(alpha_max, max(alpha_min, sksk / beta))


def spg(f, df, x0, project=None, gamma=1e-4, memory=1, sigma_one=0.1, sigma_two=0.9,
        lambda_min=1e-10, alpha0=None, alpha_min=1e-5, alpha_max=1e3,
        epsilon_one=1e-10, epsilon_two=1e-6, use_infinity_norm=True, verbose=0,
        max_iterations=10000, max_feval=1000000):
    """spg.py:46-283.  Returns (x, f, n_iter, n_feval) with the reference's
    0-based n_iter, and emits the same UserWarnings."""
    multivariate = not np.isscalar(x0)
    x = x0.copy() if multivariate else x0
    if project is not None:
        x = project(x)                                              # :148
    alpha = alpha0
    f_mem = np.zeros(memory)                                        # :153 (zeros, not NaN)
    f_old = f(x)
    n_feval = 1
    converged = False
    n_iter = -1
    for n_iter in range(max_iterations):
        x_old = x.copy() if multivariate else x
        g = df(x)                                                   # :176
        if alpha is None:                                           # :178-189
            if project is None:
                alpha = 1.0 / np.max(np.abs(g))
            else:
                alpha_inv = np.max(np.abs(project(x - g) - x))
                alpha = 1.0 / alpha_inv if abs(alpha_inv) > 1e-12 else 1.0
        d = -alpha * g
        if project is not None:
            d = project(x + d)
            d -= x
        f_mem = np.roll(f_mem, 1)
        f_mem[0] = f_old
        f_max = None
        for v in f_mem:                                             # :201-203
            if f_max is None or v >= f_max:
                f_max = v
        delta = np.sum(d * g)
        lam = 1
        x = x_old + d
        f_new = f(x)
        n_feval += 1
        while f_new > f_max + gamma * lam * delta:                  # :214
            lam = line_search_step_length(lam, delta, f_old, f_new, sigma_one, sigma_two)
            x = x_old + lam * d
            f_new = f(x)
            n_feval += 1
            if abs(lam) < lambda_min:
                warnings.warn("step size below tolerance in SPG line search", UserWarning)
                break
        y = g.copy() if multivariate else g
        g = df(x)                                                   # :233
        y = g - y
        sksk = lam ** 2 * np.sum(d * d)
        beta = lam * np.sum(d * y)
        alpha = cauchy_step_size(beta, sksk, alpha_min=alpha_min, alpha_max=alpha_max)
        f_old = f(x)                                                # :243
        n_feval += 1
        res = -g if project is None else project(x - g) - x
        res_norm = np.sum(res ** 2) ** 0.5
        converged = res_norm < epsilon_two
        if use_infinity_norm:
            converged = converged or np.max(np.abs(res)) < epsilon_one
        if converged:
            break
        if n_feval > max_feval:
            warnings.warn("maximum number of function evaluations exceeded in SPG", UserWarning)
            break
    if n_iter == max_iterations - 1 and not converged:
        warnings.warn("maximum number of iterations exceeded in SPG", UserWarning)
    return x, f_old, n_iter, n_feval


def quad_simplex_spg_py(A, b, x0, **kw):
    """spg.py:286-398, statement for statement.  Returns (x, passes, n_feval)."""
    p = qp_params(**kw)
    x = simplex_project_vector_py(x0)
    x_old = np.zeros_like(x)
    f_mem = np.full(p["memory"], np.nan)
    Ax = A.dot(x)
    f_old = 0.5 * x.dot(Ax) + x.dot(b)
    n_feval = 1
    alpha = 1.0
    passes = 0
    for n_iter in range(p["max_iterations"]):
        passes = n_iter + 1
        x_old[:] = x
        g = Ax + b
        if n_iter == 0:
            if p["alpha_min"] <= p["alpha0"] <= p["alpha_max"]:
                alpha = p["alpha0"]
            else:
                alpha_inv = np.max(np.abs(simplex_project_vector_py(x - g) - x))
                if abs(alpha_inv) < 1e-12:
                    alpha_inv = 1.0
                alpha = min(max(p["alpha_min"], 1.0 / alpha_inv), p["alpha_max"])
        d = simplex_project_vector_py(x - alpha * g) - x
        f_mem = np.roll(f_mem, 1)
        f_mem[0] = f_old
        f_max = np.nanmax(f_mem)
        delta = d.dot(g)
        lam = 1
        x = x_old + d
        Ax = A.dot(x)
        f_new = 0.5 * x.dot(Ax) + x.dot(b)
        n_feval += 1
        while f_new > f_max + p["gamma"] * lam * delta:
            lam = line_search_step_length(lam, delta, f_old, f_new,
                                          p["sigma_one"], p["sigma_two"])
            x = x_old + lam * d
            Ax = A.dot(x)
            f_new = 0.5 * x.dot(Ax) + x.dot(b)
            n_feval += 1
            if abs(lam) < p["lambda_min"]:
                break
        y = Ax + b - g
        g = y + g
        sksk = lam ** 2 * d.dot(d)
        beta = lam * d.dot(y)
        alpha = cauchy_step_size(beta, sksk, p["alpha_min"], p["alpha_max"])
        f_old = 0.5 * x.dot(Ax) + x.dot(b)
        n_feval += 1
        res = simplex_project_vector_py(x - g) - x
        res_norm = np.sum(res ** 2) ** 0.5
        if res_norm < p["epsilon_two"] or np.max(np.abs(res)) < p["epsilon_one"]:
            break
        if n_feval > p["max_feval"]:
            break
    return x, passes, n_feval


def qp_batch(A, B, Z0, b_layout, return_iters=False, **kw):
    """Solve the n per-sample QPs.  ``b_layout='kn'``: B is k x n and
    b_t = -B[:, t] (archetypal_analysis.py:359-366); ``'nk'``: B is n x k and
    b_t = -B[t] (gpnh_convex_coding.py:244-251)."""
    p = qp_params(**kw)
    A = np.ascontiguousarray(A, dtype=np.float64)
    B = np.ascontiguousarray(B, dtype=np.float64)
    Z0 = np.ascontiguousarray(Z0, dtype=np.float64)
    n, k = Z0.shape
    lib = clib()
    if lib is not None:
        Z = np.empty_like(Z0)
        iters = np.zeros(n, dtype=np.int32)
        cp = _CParams(p["gamma"], int(p["memory"]), p["sigma_one"], p["sigma_two"],
                      p["lambda_min"], p["alpha0"], p["alpha_min"], p["alpha_max"],
                      p["epsilon_one"], p["epsilon_two"],
                      int(p["max_iterations"]), int(p["max_feval"]))
        sj, st = (n, 1) if b_layout == "kn" else (1, k)
        lib.orc_qp_batch(_dptr(A), _dptr(B), sj, st, _dptr(Z0), _dptr(Z), n, k,
                         ctypes.byref(cp), iters.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    else:
        Z = np.empty_like(Z0)
        iters = np.zeros(n, dtype=np.int32)
        for t in range(n):
            b = -B[:, t] if b_layout == "kn" else -B[t]
            Z[t], iters[t], _ = quad_simplex_spg_py(A, b, Z0[t], **p)
    return (Z, iters) if return_iters else Z


# --------------------------------------------------------------------------
# stochastic matrices (stochastic_matrices.py:15-39)
# --------------------------------------------------------------------------
def _rng(random_state):
    if random_state is None:
        return np.random.mtrand._rand
    if isinstance(random_state, (int, np.integer)):
        return np.random.RandomState(random_state)
    return random_state


def right_stochastic_matrix(shape, random_state=None):
    m = _rng(random_state).uniform(size=shape)
    return m / m.sum(axis=1)[:, np.newaxis]


def left_stochastic_matrix(shape, random_state=None):
    m = _rng(random_state).uniform(size=shape)
    return m / m.sum(axis=0)[np.newaxis, :]


# --------------------------------------------------------------------------
# FurthestSum (furthest_sum.py:23-127)
# --------------------------------------------------------------------------
def furthest_sum(D, k, start, exclude=None, extra_steps=1):
    """Restatement of furthest_sum.py:23-170 on a dissimilarity matrix D.

    The candidate pool is kept as two parallel Python lists (index, running
    sum) in the reference's positional order; each pick is 'stable ascending
    sort by sum, take the last', i.e. among equal sums the one positioned last
    in the pool wins, and the pool is left sorted (furthest_sum.py:17-20)."""
    D = np.asarray(D)
    if D.shape[0] != D.shape[1]:
        raise ValueError("Dissimilarity matrix must be square, but got shape %r" % list(D.shape))
    if k == 0:
        return []
    exclude = [] if exclude is None else list(exclude)
    n = D.shape[0]
    if start >= n:
        raise ValueError("Start index %r is out of bounds (n_samples = %d)" % (start, n))
    if any(e == start for e in exclude):
        raise ValueError("Start index %r is excluded" % start)
    if len(exclude) < n and k > n - len(exclude):
        raise ValueError("Too few point available to select requested number of components "
                         "(n_components=%d, n_samples=%d, n_excluded=%d)" % (k, n, len(exclude)))
    selected = np.full((k,), start)
    banned = set(int(e) for e in exclude) | {int(start)}
    idx = [i for i in range(n) if i not in banned]
    sums = [D[i, start] for i in idx]

    def pick():
        order = sorted(range(len(idx)), key=lambda j: sums[j])      # stable
        idx[:] = [idx[j] for j in order]
        sums[:] = [sums[j] for j in order]
        sums.pop()
        return idx.pop()

    def add(new):
        for j, i in enumerate(idx):
            sums[j] += D[new, i]

    for c in range(1, k):
        selected[c] = pick()
        add(selected[c])
    for step in range(max(extra_steps, 0)):
        c = step % k
        old = selected[c]
        for j, i in enumerate(idx):
            sums[j] -= D[i, old]
        q = 0
        for s in selected:
            if s != old:
                q += D[old, s]
        idx.append(old)
        sums.append(q)
        selected[c] = pick()
        add(selected[c])
    return selected


def dissimilarities_from_kernel(K):
    """archetypal_analysis.py:95-100."""
    d = np.diag(K)
    n = K.shape[0]
    return np.sqrt(np.tile(d, (n, 1)) - 2 * K + np.tile(d[:, np.newaxis], (1, n)))


# --------------------------------------------------------------------------
# kernel-AA / AA cost, gradients, updates (archetypal_analysis.py:200-396)
# --------------------------------------------------------------------------
def kernel_aa_cost(K, Z, C, alpha):
    """archetypal_analysis.py:200-217."""
    n = K.shape[0]
    da = np.diag(alpha)
    CK = C.dot(K)
    CKCt = CK.dot(C.T)
    CKZ = CK.dot(Z)
    ZtZ = Z.T.dot(Z)
    return 0.5 * (np.trace(K) - 2 * np.trace(da.dot(CKZ))
                  + np.trace(da.dot(ZtZ.dot(da)).dot(CKCt))) / n


def update_kernel_aa_dictionary(K, C, alpha, trace_K, KZ, ZtZ, project=None, **kw):
    """archetypal_analysis.py:304-321 (f :273-281, df :284-290; both / k).
    ``project`` defaults to the fast row projection; pass
    ``simplex_project_rows_py`` for the reference's exact summation order."""
    da = np.diag(alpha)
    KZD = KZ.dot(da)
    DZtZD = da.dot(ZtZ.dot(da))

    def f(x):
        k = x.shape[0]
        return 0.5 * (trace_K - 2 * np.trace(x.dot(KZD))
                      + np.trace(DZtZD.dot(x.dot(K.dot(x.T))))) / k

    def df(x):
        return (DZtZD.dot(x.dot(K)) - KZD.T) / x.shape[0]

    return spg(f, df, C, project=project or simplex_project_rows, **kw)


def update_aa_dictionary(X, C, alpha, trace_XXt, XXtZ, ZtZ, project=None, **kw):
    """archetypal_analysis.py:324-341 (f :261-270 divides by k = C.shape[0];
    df :293-301 divides by n = C.shape[1] -- the inconsistency is the
    reference's and is reproduced)."""
    da = np.diag(alpha)
    XXtZD = XXtZ.dot(da)
    DZtZD = da.dot(ZtZ.dot(da))

    def f(x):
        CX = _rr(x, X)
        return 0.5 * (trace_XXt - 2 * np.trace(x.dot(XXtZD))
                      + np.trace(DZtZD.dot(CX.dot(CX.T)))) / x.shape[0]

    def df(x):
        CX = _rr(x, X)
        return (DZtZD.dot(CX.dot(X.T) if _OPERAND_DTYPE is None else _rl(X, CX.T).T) - XXtZD.T) / x.shape[1]

    return spg(f, df, C, project=project or simplex_project_rows, **kw)


def update_kernel_aa_weights(Z, alpha, CK, CKCt, return_iters=False, **kw):
    """archetypal_analysis.py:369-396."""
    da = np.diag(alpha)
    A = da.dot(CKCt.dot(da))
    B = da.dot(CK)
    return qp_batch(A, B, Z, "kn", return_iters=return_iters, **kw)


def _scale_objective(alpha, trace_K, CKZ, ZtZ, CKCt):
    """archetypal_analysis.py:220-229."""
    n = CKZ.shape[1]
    a2 = np.outer(alpha, alpha)
    return 0.5 * (trace_K - 2 * alpha.dot(np.diag(CKZ)) + np.sum(a2 * ZtZ * CKCt)) / n


def _scale_gradient(alpha, CKZ, ZtZ, CKCt):
    """archetypal_analysis.py:232-240."""
    n = CKZ.shape[1]
    return np.diag(ZtZ.dot(np.diag(alpha).dot(CKCt)) - CKZ) / n


def update_scale_factors(alpha, trace_K, CKZ, ZtZ, CKCt, delta, **kw):
    """archetypal_analysis.py:243-258."""
    return spg(lambda a: _scale_objective(a, trace_K, CKZ, ZtZ, CKCt),
               lambda a: _scale_gradient(a, CKZ, ZtZ, CKCt),
               alpha,
               project=lambda a: np.fmin(np.fmax(1.0 - delta, a), 1.0 + delta),
               **kw)[0]


def _cost_increased(old, new, tol, stage, require):
    """archetypal_analysis.py:167-174."""
    if (new > old) and (abs(new - old) > tol) and require:
        raise RuntimeError("factorization cost increased after {} update".format(stage))


def _converged(criterion):
    """archetypal_analysis.py:177-197."""
    if criterion == "abs_delta_f":
        return lambda old, new, tol: abs(new - old) < tol
    if criterion == "rel_delta_f":
        return lambda old, new, tol: abs((new - old) / max(abs(new), abs(old))) < tol
    raise ValueError("unsupported stopping criterion '%s'" % criterion)


def _half_tr(trace_data, trace_cross, trace_quad, n):
    return 0.5 * (trace_data - 2 * trace_cross + trace_quad) / n


def iterate_kernel_aa(K, Z, C, alpha, delta=0, update_weights=True, update_dictionary=True,
                      update_scale_factors=True, tolerance=1e-6, max_iterations=1000,
                      **kwargs):
    """archetypal_analysis.py:399-531.  Returns the reference's 7-tuple."""
    n, k = Z.shape
    require = kwargs.get("require_monotonic_cost_decrease", True)
    done = _converged(kwargs.get("stopping_criterion", "abs_delta_f"))
    dkw = kwargs.get("dictionary_solver_kwargs", {})
    wkw = kwargs.get("weights_solver_kwargs", {})
    skw = kwargs.get("scale_factors_solver_kwargs", {})

    da = np.diag(alpha)
    ZtZ = Z.T.dot(Z)
    CK = C.dot(K)
    CKCt = CK.dot(C.T)
    KZ = K.dot(Z)
    CKZ = C.dot(KZ)
    trK = K.trace()

    def cost():
        return _half_tr(trK, da.dot(CKZ).trace(), da.dot(ZtZ.dot(da)).dot(CKCt).trace(), n)

    new = cost()
    times, deltas = [], []
    n_iter = -1
    for n_iter in range(max_iterations):
        t0 = time.perf_counter()
        old = new
        if update_scale_factors and delta != 0:
            alpha = update_scale_factors_fn(alpha, trK, CKZ, ZtZ, CKCt, delta, **skw)
            da = np.diag(alpha)
            new = cost()
            _cost_increased(old, new, tolerance, "scale factors", require)
        if update_dictionary:
            C = update_kernel_aa_dictionary(K, C, alpha, trK, KZ, ZtZ, **dkw)[0]
            CK = C.dot(K)
            CKCt = CK.dot(C.T)
            CKZ = C.dot(KZ)
            new = cost()
            _cost_increased(old, new, tolerance, "dictionary", require)
        if update_weights:
            Z = update_kernel_aa_weights(Z, alpha, CK, CKCt, **wkw)
            ZtZ = Z.T.dot(Z)
            KZ = K.dot(Z)
            CKZ = C.dot(KZ)
            new = cost()
            _cost_increased(old, new, tolerance, "weights", require)
        times.append(time.perf_counter() - t0)
        deltas.append(new - old)
        if done(old, new, tolerance):
            break
    return Z, C, alpha, new, n_iter, np.mean(times), deltas


update_scale_factors_fn = update_scale_factors


def iterate_aa(X, Z, C, alpha, delta=0, update_weights=True, update_dictionary=True,
               update_scale_factors=True, tolerance=1e-6, max_iterations=1000,
               trace_XXt=None, timings=None, cost_log=None, **kwargs):
    """archetypal_analysis.py:534-670: the reference's op sequence -- with
    ``dictionary_solver_kwargs={'max_iterations': 1}`` 11 GEMM passes over X per
    outer iteration (7 inside spg, 2 at :618-619, 2 at :641-642).

    ``trace_XXt``: the reference forms np.trace(X.dot(X.T)) (:552, an n x n
    temporary); pass the value (= ||X||_F^2) to avoid that at large n.
    ``timings``: optional dict accumulating seconds per phase
    ('dictionary', 'gram', 'weights').
    ``cost_log``: optional list that receives (stage, cost) after every update
    (the values the reference's monotonicity checks see, :612,:630,:651)."""
    n, k = Z.shape
    require = kwargs.get("require_monotonic_cost_decrease", True)
    done = _converged(kwargs.get("stopping_criterion", "abs_delta_f"))
    dkw = kwargs.get("dictionary_solver_kwargs", {})
    wkw = kwargs.get("weights_solver_kwargs", {})
    skw = kwargs.get("scale_factors_solver_kwargs", {})

    def tick(name, t0):
        if timings is not None:
            timings[name] = timings.get(name, 0.0) + time.perf_counter() - t0

    da = np.diag(alpha)
    ZtZ = Z.T.dot(Z)
    CX = _rr(C, X)
    CXXt = CX.dot(X.T) if _OPERAND_DTYPE is None else _rl(X, CX.T).T
    CXXtCt = CX.dot(CX.T)
    XtZ = X.T.dot(Z) if _OPERAND_DTYPE is None else _rr(Z.T, X).T
    XXtZ = _rl(X, XtZ)
    CXXtZ = C.dot(XXtZ)
    trX = np.trace(X.dot(X.T)) if trace_XXt is None else trace_XXt

    def cost():
        return _half_tr(trX, da.dot(CXXtZ).trace(), da.dot(ZtZ.dot(da)).dot(CXXtCt).trace(), n)

    new = cost()
    times, deltas = [], []
    n_iter = -1
    for n_iter in range(max_iterations):
        t0 = time.perf_counter()
        old = new
        if update_scale_factors and delta != 0:
            alpha = update_scale_factors_fn(alpha, trX, CXXtZ, ZtZ, CXXtCt, delta, **skw)
            da = np.diag(alpha)
            new = cost()
            if cost_log is not None:
                cost_log.append(("scale factors", new))
            _cost_increased(old, new, tolerance, "scale factors", require)
        if update_dictionary:
            t1 = time.perf_counter()
            C = update_aa_dictionary(X, C, alpha, trX, XXtZ, ZtZ, **dkw)[0]
            tick("dictionary", t1)
            t1 = time.perf_counter()
            CX = _rr(C, X)
            CXXt = CX.dot(X.T) if _OPERAND_DTYPE is None else _rl(X, CX.T).T
            CXXtCt = CX.dot(CX.T)
            CXXtZ = C.dot(XXtZ)
            tick("gram", t1)
            new = cost()
            if cost_log is not None:
                cost_log.append(("dictionary", new))
            _cost_increased(old, new, tolerance, "dictionary", require)
        if update_weights:
            t1 = time.perf_counter()
            Z = update_kernel_aa_weights(Z, alpha, CXXt, CXXtCt, **wkw)
            tick("weights", t1)
            t1 = time.perf_counter()
            ZtZ = Z.T.dot(Z)
            XtZ = X.T.dot(Z) if _OPERAND_DTYPE is None else _rr(Z.T, X).T
            XXtZ = _rl(X, XtZ)
            CXXtZ = C.dot(XXtZ)
            tick("gram", t1)
            new = cost()
            if cost_log is not None:
                cost_log.append(("weights", new))
            _cost_increased(old, new, tolerance, "weights", require)
        times.append(time.perf_counter() - t0)
        deltas.append(new - old)
        if done(old, new, tolerance):
            break
    return Z, C, alpha, new, n_iter, np.mean(times), deltas


# --------------------------------------------------------------------------
# initialisation (archetypal_analysis.py:51-164)
# --------------------------------------------------------------------------
def init_kernel_aa(K, k, init="furthest_sum", random_state=None, **kw):
    """RNG order of archetypal_analysis.py:151-164: dictionary first, then weights."""
    rng = _rng(random_state)
    n = K.shape[0]
    if init is None:
        init = "furthest_sum"
    if init == "furthest_sum":
        start = kw.get("start_index", None)
        if start is None:
            start = rng.randint(n)
        sel = furthest_sum(dissimilarities_from_kernel(K), k, start,
                           kw.get("exclude", None) if kw.get("exclude", None) is not None
                           else np.array([], dtype="i8"),
                           kw.get("n_extra_steps", 10))
        C = np.zeros((k, n), dtype=K.dtype)
        for i in range(k):
            C[i, sel[i]] = 1
    elif init == "random":
        C = right_stochastic_matrix((k, n), random_state=rng)
    else:
        raise ValueError("Invalid init parameter: got %r" % (init,))
    Z = right_stochastic_matrix((n, k), random_state=rng)
    return C, Z


def init_scale_factors(k, delta=0, random_state=None):
    """archetypal_analysis.py:73-81."""
    rng = _rng(random_state)
    if delta != 0:
        return rng.uniform(low=(1 - delta), high=(1 + delta), size=(k,))
    return np.ones(k)


def archetypal_analysis(X, k, delta=0, init=None, tolerance=1e-6, max_iterations=1000,
                        random_state=None, dictionary=None, weights=None, alpha=None, **kwargs):
    """ArchetypalAnalysis.fit_transform (archetypal_analysis.py:1026-1149) as a
    function.  Returns dict(weights, dictionary, archetypes, alpha, cost, n_iter,
    cost_deltas)."""
    rng = _rng(random_state)
    X = np.asarray(X, dtype=np.float64)
    if init == "custom":
        C, Z = dictionary, weights
    else:
        C, Z = init_kernel_aa(X.dot(X.T), k, init=init, random_state=rng)
    if alpha is None:
        alpha = init_scale_factors(k, delta, rng)
    Z, C, alpha, cost, n_iter, _, deltas = iterate_aa(
        X, Z.copy(), C.copy(), alpha.copy(), delta=delta, tolerance=tolerance,
        max_iterations=max_iterations, **kwargs)
    if delta != 0:
        C = np.diag(alpha).dot(C)
    return dict(weights=Z, dictionary=C, archetypes=C.dot(X), alpha=alpha, cost=cost,
                n_iter=n_iter, cost_deltas=deltas)


def kernel_aa(K, k, delta=0, init=None, tolerance=1e-6, max_iterations=1000,
              random_state=None, dictionary=None, weights=None, alpha=None, **kwargs):
    """KernelAA.fit_transform (archetypal_analysis.py:773-895) as a function."""
    rng = _rng(random_state)
    K = np.asarray(K, dtype=np.float64)
    if init == "custom":
        C, Z = dictionary, weights
    else:
        C, Z = init_kernel_aa(K, k, init=init, random_state=rng)
    if alpha is None:
        alpha = init_scale_factors(k, delta, rng)
    Z, C, alpha, cost, n_iter, _, deltas = iterate_kernel_aa(
        K, Z.copy(), C.copy(), alpha.copy(), delta=delta, tolerance=tolerance,
        max_iterations=max_iterations, **kwargs)
    return dict(weights=Z, dictionary=C, alpha=alpha, cost=cost, n_iter=n_iter,
                cost_deltas=deltas)


def aa_transform(archetypes, X_new, k, max_iterations, random_state=None, **wkw):
    """ArchetypalAnalysis.transform (archetypal_analysis.py:1151-1199)."""
    rng = _rng(random_state)
    m = X_new.shape[0]
    A = archetypes.dot(archetypes.T)
    B = archetypes.dot(X_new.T)
    Z0 = right_stochastic_matrix((m, k), random_state=rng)
    kw = dict(wkw)
    kw["max_iterations"] = max_iterations      # :1194 passes self.max_iterations
    Z = qp_batch(A, B, Z0, "kn", **kw)
    cost = 0.5 * np.linalg.norm(X_new - Z.dot(archetypes)) ** 2 / m
    return Z, cost


# --------------------------------------------------------------------------
# GPNH convex coding (gpnh_convex_coding.py)
# --------------------------------------------------------------------------
def gpnh_regularization(W):
    """gpnh_convex_coding.py:179-196."""
    p, k = W.shape
    if k == 1:
        return 0.0
    phi = 0.0
    for i in range(k):
        for j in range(i + 1, k):
            phi += np.linalg.norm(W[:, i] - W[:, j]) ** 2
    return 2.0 / (k * p * (k - 1.0)) * phi


def gpnh_cost(X, Z, W, lambda_W=0):
    """gpnh_convex_coding.py:199-210."""
    c = 0.5 * np.linalg.norm(X - Z.dot(W.T)) ** 2 / X.shape[0]
    if lambda_W != 0:
        c += lambda_W * gpnh_regularization(W)
    return c


def gpnh_gw(p, k):
    """gpnh_convex_coding.py:296-300."""
    if k > 1:
        return (4.0 / (p * k * (k - 1))) * (k * np.eye(k) - 1)
    return np.zeros((k, k))


def update_gpnh_dictionary(X, Z, ZtZ, GW, lambda_W=0):
    """gpnh_convex_coding.py:213-226."""
    n = X.shape[0]
    lhs = ZtZ / n + lambda_W * GW
    rhs = _rr(Z.T, X) / n
    return np.linalg.lstsq(lhs, rhs, rcond=None)[0].T


def update_gpnh_weights(X, Z, W, return_iters=False, **kw):
    """gpnh_convex_coding.py:254-279."""
    return qp_batch(W.T.dot(W), _rl(X, W), Z, "nk", return_iters=return_iters, **kw)


def iterate_gpnh(X, Z, W, lambda_W=0, update_weights=True, update_dictionary=True,
                 tolerance=1e-6, max_iterations=1000, **kwargs):
    """gpnh_convex_coding.py:282-402.  Returns the reference's 6-tuple."""
    p = X.shape[1]
    n, k = Z.shape
    require = kwargs.get("require_monotonic_cost_decrease", True)
    done = _converged(kwargs.get("stopping_criterion", "abs_delta_f"))
    wkw = kwargs.get("weights_solver_kwargs", {})
    WtXt = W.T.dot(X.T)
    ZtZ = Z.T.dot(Z)
    WtW = W.T.dot(W)
    GW = gpnh_gw(p, k)
    trX = X.T.dot(X).trace()
    pen = lambda_W * gpnh_regularization(W) if lambda_W != 0 else 0

    def cost():
        return 0.5 * (trX - 2 * WtXt.dot(Z).trace() + ZtZ.dot(WtW).trace()) / n + pen

    new = cost()
    times, deltas = [], []
    n_iter = -1
    for n_iter in range(max_iterations):
        t0 = time.perf_counter()
        old = new
        if update_dictionary:
            W = update_gpnh_dictionary(X, Z, ZtZ, GW, lambda_W=lambda_W)
            WtXt = W.T.dot(X.T)
            WtW = W.T.dot(W)
            pen = lambda_W * gpnh_regularization(W) if lambda_W != 0 else 0
            new = cost()
            _cost_increased(old, new, tolerance, "dictionary", require)
        if update_weights:
            Z = update_gpnh_weights(X, Z, W, **wkw)
            ZtZ = Z.T.dot(Z)
            new = cost()
            _cost_increased(old, new, tolerance, "weights", require)
        times.append(time.perf_counter() - t0)
        deltas.append(new - old)
        if done(old, new, tolerance):
            break
    return Z, W, new, n_iter, np.mean(times), deltas


def init_gpnh(X, k, init="random", random_state=None, **kw):
    """gpnh_convex_coding.py:41-143 (dictionary first, then weights)."""
    rng = _rng(random_state)
    n, p = X.shape
    if init is None:
        init = "random"
    if init == "random":
        W = np.sqrt(np.abs(X).mean() / k) * rng.randn(p, k)
    elif init == "furthest_sum":
        K = X.dot(X.T)
        start = kw.get("start_index", None)
        if start is None:
            start = rng.randint(n)
        sel = furthest_sum(dissimilarities_from_kernel(K), k, start,
                           np.array([], dtype="i8"), kw.get("n_extra_steps", 10))
        W = np.zeros((p, k), dtype=K.dtype)
        for i in range(k):
            W[:, i] = X[sel[i]]
    else:
        raise ValueError("Invalid init parameter: got %r" % (init,))
    Z = right_stochastic_matrix((n, k), random_state=rng)
    return W, Z


def gpnh_convex_coding(X, k, lambda_W=0, init=None, tolerance=1e-6, max_iterations=1000,
                       random_state=None, dictionary=None, weights=None, **kwargs):
    """GPNHConvexCoding.fit_transform (gpnh_convex_coding.py:501-606) as a function."""
    rng = _rng(random_state)
    X = np.asarray(X, dtype=np.float64)
    if init == "custom":
        W, Z = dictionary, weights
    else:
        W, Z = init_gpnh(X, k, init=init, random_state=rng)
    Z, W, cost, n_iter, _, deltas = iterate_gpnh(
        X, Z.copy(), W.copy(), lambda_W=lambda_W, tolerance=tolerance,
        max_iterations=max_iterations, **kwargs)
    return dict(weights=Z, dictionary=W, cost=cost, n_iter=n_iter, cost_deltas=deltas)


def gpnh_transform(W, X_new, lambda_W=0, tolerance=1e-6, max_iterations=1000, random_state=None,
                   **kwargs):
    """GPNHConvexCoding.transform (gpnh_convex_coding.py:623-652): fresh random weights from the
    estimator's generator (:531-537), then the loop of :282-402 with the dictionary held fixed.
    Returns (weights, cost)."""
    rng = _rng(random_state)
    X_new = np.asarray(X_new, dtype=np.float64)
    k = W.shape[1]
    Z0 = right_stochastic_matrix((X_new.shape[0], k), random_state=rng)
    Z, _, cost, _, _, _ = iterate_gpnh(X_new, Z0, np.array(W, dtype=np.float64), lambda_W=lambda_W,
                                       update_dictionary=False, update_weights=True,
                                       tolerance=tolerance, max_iterations=max_iterations, **kwargs)
    return Z, cost

