"""Spectral projected gradient solvers.

``spg`` mirrors reference src/convex_dim_red/spg.py:46-283: it is the GENERIC driver
for arbitrary Python callables ``f``, ``df``, ``project`` (the reference's own tests
call it on scalars), so it is host control flow by nature; the callables do the
arithmetic.  The two hot uses of SPG in the package do not go through this function:
the dictionary update runs its SPG on the device (``_backend.Context.dictionary_update``
-> csrc/solver.hip) and the per-sample simplex QPs run in ``k_qp`` (csrc/kernels_qp.hip),
which is what ``quad_simplex_spg`` below calls for a single sample.
"""
import time
import warnings

import numpy as np

from . import _backend


def spg_line_search_step_length(current_step_length, delta, f_old, f_new,
                                sigma_one=0.1, sigma_two=0.9):
    """Safeguarded quadratic interpolation (reference spg.py:19-33; sigma_one is an
    absolute lower bound there, kept)."""
    lam = current_step_length
    with np.errstate(divide="ignore", invalid="ignore"):
        candidate = -0.5 * lam ** 2 * delta / (f_new - f_old - lam * delta)
    accept = sigma_one <= candidate <= sigma_two * lam
    return candidate if accept else 0.5 * lam


def spg_line_search_cauchy_step_size(beta, sksk, alpha_min=1e-3, alpha_max=1e3):
    """Barzilai-Borwein step with clamping (reference spg.py:36-43)."""
    if beta <= 0:
        return alpha_max
    return min(alpha_max, max(alpha_min, sksk / beta))


class _Report(object):
    """verbose table of the reference (spg.py:159-164,256-269)."""
    ROW = "{:12d} | {:12d} | {: 12.6e} | {: 12.6e} | {: 12.6e}"

    def __init__(self, on):
        self.on = on

    def header(self, n_feval, f0):
        if self.on:
            print("{:<12s} | {:<12s} | {:<13s} | {:<13s} | {:<12s}".format(
                "n_iter", "n_feval", "f", "conv_crit", "time"))
            print("-" * 79)
            print(self.ROW.format(0, n_feval, f0, -1, 0))

    def row(self, it, n_feval, f, crit, dt):
        if self.on:
            print(self.ROW.format(it, n_feval, f, crit, dt))

    def converged(self, it):
        if self.on:
            print("-" * 79)
            print("*** Converged at iteration {:d} ***".format(it))


def spg(f, df, x0, project=None, gamma=1e-4, memory=1,
        sigma_one=0.1, sigma_two=0.9, lambda_min=1e-10,
        alpha0=None, alpha_min=1e-5, alpha_max=1e3,
        epsilon_one=1e-10, epsilon_two=1e-6,
        use_infinity_norm=True, verbose=0,
        max_iterations=10000, max_feval=1000000):
    """Non-monotone spectral projected gradient (Birgin, Martinez, Raydan; reference
    spg.py:46-283).  Returns ``(x, f(x), n_iter, n_feval)`` with the reference's
    0-based ``n_iter`` and the same UserWarnings."""
    array_like = not np.isscalar(x0)

    def snapshot(v):
        return v.copy() if array_like else v

    report = _Report(verbose)
    x = snapshot(x0)
    if project is not None:
        x = project(x)
    step = alpha0                      # BB step; None => derived on the first pass
    history = np.zeros(memory)         # reference initialises with zeros (spg.py:153)
    f_cur = f(x)
    evals = 1
    report.header(evals, f_cur)

    done = False
    it = -1
    for it in range(max_iterations):
        t0 = time.perf_counter()
        x_prev = snapshot(x)
        grad = df(x)
        if step is None:
            if project is None:
                step = 1.0 / np.max(np.abs(grad))
            else:
                reach = np.max(np.abs(project(x - grad) - x))
                step = 1.0 / reach if abs(reach) > 1e-12 else 1.0

        direction = -step * grad
        if project is not None:
            direction = project(x + direction)
            direction -= x

        history = np.roll(history, 1)
        history[0] = f_cur
        f_ref = None
        for past in history:
            if f_ref is None or past >= f_ref:
                f_ref = past

        slope = np.sum(direction * grad)
        lam = 1
        x = x_prev + direction
        f_try = f(x)
        evals += 1
        while f_try > f_ref + gamma * lam * slope:
            lam = spg_line_search_step_length(lam, slope, f_cur, f_try,
                                              sigma_one=sigma_one, sigma_two=sigma_two)
            x = x_prev + lam * direction
            f_try = f(x)
            evals += 1
            if abs(lam) < lambda_min:
                warnings.warn("step size below tolerance in SPG line search", UserWarning)
                break

        grad_prev = snapshot(grad)
        grad = df(x)
        grad_change = grad - grad_prev
        sksk = lam ** 2 * np.sum(direction * direction)
        beta = lam * np.sum(direction * grad_change)
        step = spg_line_search_cauchy_step_size(beta, sksk, alpha_min=alpha_min,
                                                alpha_max=alpha_max)
        f_cur = f(x)
        evals += 1

        residual = -grad if project is None else project(x - grad) - x
        res_norm = np.sum(residual ** 2) ** 0.5
        report.row(it + 1, evals, f_cur, res_norm, time.perf_counter() - t0)

        done = res_norm < epsilon_two
        if use_infinity_norm:
            done = done or np.max(np.abs(residual)) < epsilon_one
        if done:
            report.converged(it + 1)
            break
        if evals > max_feval:
            warnings.warn("maximum number of function evaluations exceeded in SPG", UserWarning)
            break

    if it == max_iterations - 1 and not done:
        warnings.warn("maximum number of iterations exceeded in SPG", UserWarning)
    return x, f_cur, it, evals


def quad_simplex_spg(A, b, x0, gamma=1e-4, memory=1,
                     sigma_one=0.1, sigma_two=0.9, lambda_min=1e-10,
                     alpha0=-1.0, alpha_min=1e-5, alpha_max=1e3,
                     epsilon_one=1e-10, epsilon_two=1e-6,
                     max_iterations=1000, max_feval=2000):
    """min 0.5 x'Ax + b'x on the unit simplex for ONE sample (reference
    spg.py:286-398), solved by the batched device kernel with a batch of one."""
    A = np.asarray(A, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    x0 = np.asarray(x0, dtype=np.float64)
    Z = _backend.qp_batch(A, -b.reshape(1, -1), x0.reshape(1, -1), "nk",
                          gamma=gamma, memory=memory, sigma_one=sigma_one, sigma_two=sigma_two,
                          lambda_min=lambda_min, alpha0=alpha0, alpha_min=alpha_min,
                          alpha_max=alpha_max, epsilon_one=epsilon_one, epsilon_two=epsilon_two,
                          max_iterations=max_iterations, max_feval=max_feval)
    return Z[0]
