"""Archetypal analysis on MI355X.

Mirrors the public classes and the test-visible private functions of reference
src/convex_dim_red/archetypal_analysis.py (same names, argument meaning, return
tuples, exceptions and warnings); every numeric step runs on the GPU through
``_backend.Context`` (csrc/solver.hip) -- the alternating loop included: the monotonicity
check, the stopping rule and the k-vector scale-factor update (``delta != 0``) run on the
device (aa_iterate), the host reads one status record per batch of iterations.  What stays
on the host is argument validation, RNG-ordered initialisation (FurthestSum's candidate
list), warnings, timing and verbose tables; the host-driven one-update-at-a-time loop is
kept as a cross-check (``_DEVICE_LOOP = False``).

Unlike the reference, ``ArchetypalAnalysis`` never forms the n x n kernel
``data.dot(data.T)`` (reference :1032), the n x n dissimilarity matrix (:95-100) or
``np.trace(X.dot(X.T))`` (:552): FurthestSum pulls distance columns from the device
and the trace is ||X||_F^2.
"""
from __future__ import absolute_import, division, print_function

import numbers
import time
import warnings

import numpy as np
from sklearn.utils import check_array, check_random_state

from . import _backend
from .furthest_sum import furthest_sum, furthest_sum_from_columns
from .preprocessing import DeviceData
from .simplex_projection import simplex_project_rows  # noqa: F401  (reference re-export)
from .spg import quad_simplex_spg, spg  # noqa: F401
from .stochastic_matrices import right_stochastic_matrix
from .validation_utils import check_array_shape, check_stochastic_matrix

INTEGER_TYPES = (numbers.Integral, np.integer)
INITIALIZATION_METHODS = (None, 'random', 'furthest_sum',)
_DEVICE_LOOP_BATCH = 8          # outer iterations between two host polls of the device loop
_DEVICE_LOOP = True


# ----------------------------------------------------------------------------
# validation / initialisation (reference :27-164) -- host side
# ----------------------------------------------------------------------------
def _check_init_weights(weights, shape, whom):
    check_stochastic_matrix(check_array(weights), shape, whom, axis=1)


def _check_init_dictionary(dictionary, shape, whom):
    check_stochastic_matrix(check_array(dictionary), shape, whom, axis=1)


def _check_init_scale_factors(alpha, delta, shape, whom):
    check_array_shape(alpha, shape, whom)
    if np.any(np.logical_or(alpha < 1 - delta, alpha > 1 + delta)):
        raise ValueError('Initial scale factors infeasible in %s' % (whom))


def _initialize_kernel_aa_scale_factors_random(n_components, delta=0, random_state=None):
    rng = check_random_state(random_state)
    if delta != 0:
        return rng.uniform(low=(1 - delta), high=(1 + delta), size=(n_components,))
    return np.ones(n_components)


def _one_hot_rows(selected, n_samples, dtype):
    dictionary = np.zeros((len(selected), n_samples), dtype=dtype)
    dictionary[np.arange(len(selected)), np.asarray(selected, dtype=np.int64)] = 1
    return dictionary


def _initialize_kernel_aa_dictionary_furthest_sum(kernel, n_components, start_index=None,
                                                  n_extra_steps=10, exclude=None,
                                                  random_state=None):
    """Reference :84-110 for a given kernel matrix (KernelAA path)."""
    rng = check_random_state(random_state)
    n_samples = kernel.shape[0]
    if start_index is None:
        start_index = rng.randint(n_samples)
    if exclude is None:
        exclude = np.array([], dtype='i8')
    diag = np.diag(kernel)
    dissimilarities = np.sqrt(diag[np.newaxis, :] - 2 * kernel + diag[:, np.newaxis])
    selected = furthest_sum(dissimilarities, n_components, start_index, exclude, n_extra_steps)
    return _one_hot_rows(selected, n_samples, kernel.dtype)


_FURTHEST_SUM_ON_DEVICE = True       # False: always the host-driven selection (one distance column per pick)


def _furthest_sum_on_device(ctx, n_samples, n_components, start_index, n_extra_steps, exclude, cache=None):
    """FurthestSum with distance columns computed from the resident data matrix (``cache``: columns
    already fetched for the same matrix -- restarts.fit_restarts shares one over its restarts)."""
    shared = cache is not None                   # restarts.fit_restarts: one cache for the draws of all restarts
    cache = {} if cache is None else cache
    # A restart loop on a matrix whose distance columns are expensive (short and wide: the
    # HadISST shape, 322 MB per column) keeps the host-driven form: its picks end at the same few
    # extreme points restart after restart, so nearly every column comes out of the shared cache.
    column_bytes = float(getattr(ctx, 'n', 0)) * float(getattr(ctx, 'p', 0)) * (4 if ctx.dtype_code == _backend.AA_F32 else 8)
    if _FURTHEST_SUM_ON_DEVICE and not (shared and column_bytes > 64e6):
        # the whole selection in one chain of device launches (aa_furthest_sum); the host's list logic
        # below takes over when a pick meets a shared maximum (the reference's tie rule lives there)
        from .furthest_sum import _validate
        ex = [] if exclude is None else [int(e) for e in exclude]
        _validate(n_samples, n_components, start_index, ex)
        picked = ctx.furthest_sum(n_components, start_index, ex, n_extra_steps) if n_components > 0 else None
        if picked is not None:
            return picked

    def column_of(j, sense):
        j = int(j)
        if j not in cache:
            cache[j] = ctx.distance_column(j)
        return cache[j]

    def row_entry(i, j):
        return column_of(j, "into")[int(i)]

    return furthest_sum_from_columns(column_of, row_entry, n_samples, n_components,
                                     start_index, exclude=exclude, extra_steps=n_extra_steps)


def _initialize_kernel_aa_dictionary(kernel, n_components, init='furthest_sum',
                                     random_state=None, **kwargs):
    if init is None:
        init = 'furthest_sum'
    if init == 'furthest_sum':
        return _initialize_kernel_aa_dictionary_furthest_sum(
            kernel, n_components, start_index=kwargs.get('start_index', None),
            n_extra_steps=kwargs.get('n_extra_steps', 10), exclude=kwargs.get('exclude', None),
            random_state=random_state)
    if init == 'random':
        rng = check_random_state(random_state)
        return right_stochastic_matrix((n_components, kernel.shape[0]), random_state=rng)
    raise ValueError('Invalid init parameter: got %r instead of one of %r'
                     % (init, INITIALIZATION_METHODS))


def _initialize_kernel_aa_weights(kernel, n_components, init='furthest_sum', random_state=None):
    if init is None:
        init = 'furthest_sum'
    if init in ('furthest_sum', 'random'):
        rng = check_random_state(random_state)
        return right_stochastic_matrix((kernel.shape[0], n_components), random_state=rng)
    raise ValueError('Invalid init parameter: got %r instead of one of %r'
                     % (init, INITIALIZATION_METHODS))


def _initialize_kernel_aa(kernel, n_components, init='furthest_sum', random_state=None, **kwargs):
    """Dictionary first, then weights, from the same generator (reference :151-164)."""
    rng = check_random_state(random_state)
    dictionary = _initialize_kernel_aa_dictionary(kernel, n_components, init=init,
                                                  random_state=rng, **kwargs)
    weights = _initialize_kernel_aa_weights(kernel, n_components, init=init, random_state=rng)
    return dictionary, weights


class _ShapeOnly(object):
    """Stands in for the n x n kernel where only ``shape[0]`` is consulted."""

    def __init__(self, n):
        self.shape = (n, n)


def _check_if_cost_increased(old, new, tolerance, stage, require_decrease=True):
    """Reference :167-174."""
    if (new > old) and (abs(new - old) > tolerance) and require_decrease:
        raise RuntimeError('factorization cost increased after {} update'.format(stage))


def _get_stopping_criteria(stopping_criterion):
    """Reference :177-197."""
    if stopping_criterion == 'abs_delta_f':
        return lambda old_cost, new_cost, tolerance: abs(new_cost - old_cost) < tolerance
    if stopping_criterion == 'rel_delta_f':
        return lambda old_cost, new_cost, tolerance: (
            abs((new_cost - old_cost) / max(abs(new_cost), abs(old_cost))) < tolerance)
    raise ValueError("unsupported stopping criterion '%s'" % stopping_criterion)


# ----------------------------------------------------------------------------
# scale factors (reference :220-258): k-vector problem, stays on the host
# ----------------------------------------------------------------------------
def _kernel_aa_scale_factors_objective(alpha, trace_K, CKZ, ZtZ, CKCt):
    n_samples = CKZ.shape[1]
    return 0.5 * (trace_K - 2 * alpha.dot(np.diag(CKZ))
                  + np.sum(np.outer(alpha, alpha) * ZtZ * CKCt)) / n_samples


def _kernel_aa_scale_factors_gradient(alpha, CKZ, ZtZ, CKCt):
    n_samples = CKZ.shape[1]
    return np.diag(ZtZ.dot(np.diag(alpha).dot(CKCt)) - CKZ) / n_samples


def _update_kernel_aa_scale_factors(alpha, trace_K, CKZ, ZtZ, CKCt, delta, **kwargs):
    lo, hi = 1.0 - delta, 1.0 + delta
    return spg(lambda a: _kernel_aa_scale_factors_objective(a, trace_K, CKZ, ZtZ, CKCt),
               lambda a: _kernel_aa_scale_factors_gradient(a, CKZ, ZtZ, CKCt),
               alpha, project=lambda a: np.fmin(np.fmax(lo, a), hi), **kwargs)[0]


# ----------------------------------------------------------------------------
# device-backed building blocks
# ----------------------------------------------------------------------------
def _warn_from_spg_flags(stats):
    """Re-issue the UserWarnings spg() raises (reference spg.py:225-229,272-281); ``stats``
    is one update's SPGStats or the device loop's IterStats (flags OR-ed over its iterations)."""
    flags = stats.flags if hasattr(stats, 'flags') else stats.spg_flags
    if flags & _backend.SPG_FLAG_LAMBDA_MIN:
        warnings.warn('step size below tolerance in SPG line search', UserWarning)
    if flags & _backend.SPG_FLAG_MAX_FEVAL:
        warnings.warn('maximum number of function evaluations exceeded in SPG', UserWarning)
    if flags & _backend.SPG_FLAG_MAX_ITER:
        warnings.warn('maximum number of iterations exceeded in SPG', UserWarning)
    if flags & _backend.SPG_FLAG_PROJ_UNCONV:
        warnings.warn('simplex projection pass cap reached on the device', RuntimeWarning)


def _scaled_cost(ctx, alpha):
    """0.5 (tr - 2 tr(D CKZ) + tr(D ZtZ D CKCt)) / n from the device Gram products."""
    ctx.set_alpha(alpha)
    return ctx.cost()


def _kernel_aa_cost(K, weights, dictionary, alpha):
    """Kernel AA cost (reference :200-217)."""
    K = np.asarray(K, dtype=np.float64)
    with _backend.Context(dtype=np.float64) as ctx:
        ctx.set_data(K, form=_backend.FORM_KERNEL)
        ctx.set_state(dictionary, weights, alpha)
        return ctx.prepare()


def _update_dictionary_on_device(matrix, form, dictionary, alpha, trace, KZ, ZtZ, dtype, **kwargs):
    with _backend.Context(dtype=dtype) as ctx:
        ctx.set_data(matrix, form=form)
        n, k = dictionary.shape[1], dictionary.shape[0]
        placeholder = np.full((n, k), 1.0 / k)
        ctx.set_state(dictionary, placeholder, alpha)
        ctx.set_dictionary_inputs(KZ, ZtZ, trace)
        stats = ctx.dictionary_update(**kwargs)
        _warn_from_spg_flags(stats)
        return ctx.get_state()[0]


def _update_kernel_aa_dictionary(K, dictionary, alpha, trace_K, KZ, ZtZ, **kwargs):
    """SPG update of the dictionary for kernel AA (reference :304-321)."""
    return _update_dictionary_on_device(np.asarray(K, dtype=np.float64), _backend.FORM_KERNEL,
                                        dictionary, alpha, trace_K, KZ, ZtZ, np.float64, **kwargs)


def _update_aa_dictionary(X, dictionary, alpha, trace_XXt, XXtZ, ZtZ, **kwargs):
    """SPG update of the dictionary for AA (reference :324-341)."""
    return _update_dictionary_on_device(np.asarray(X), _backend.FORM_DATA, dictionary, alpha,
                                        trace_XXt, XXtZ, ZtZ, None, **kwargs)


def _update_kernel_aa_weights(weights, alpha, CK, CKCt, **solver_kwargs):
    """Per-sample simplex QPs (reference :369-396)."""
    da = np.diag(alpha)
    return _backend.qp_batch(da.dot(CKCt.dot(da)), da.dot(CK), weights, "kn", **solver_kwargs)


def _gu_update_kernel_aa_weights(CKCt, CK, initial_weights, gamma, memory, sigma_one, sigma_two,
                                 lambda_min, alpha0, alpha_min, alpha_max, epsilon_one,
                                 epsilon_two, max_iterations, max_feval):
    """Positional form of the reference gufunc (:344-366)."""
    return _backend.qp_batch(CKCt, CK, initial_weights, "kn", gamma=gamma, memory=memory,
                             sigma_one=sigma_one, sigma_two=sigma_two, lambda_min=lambda_min,
                             alpha0=alpha0, alpha_min=alpha_min, alpha_max=alpha_max,
                             epsilon_one=epsilon_one, epsilon_two=epsilon_two,
                             max_iterations=max_iterations, max_feval=max_feval)


def _iterate_on_device(ctx, label, weights, dictionary, alpha, delta, update_weights,
                       update_dictionary, update_scale_factors, tolerance, max_iterations,
                       verbose, **kwargs):
    """Alternating minimisation loop shared by both forms (reference :399-531, :534-670).
    ``ctx`` already holds the data/kernel matrix."""
    n_components = weights.shape[1]
    require_monotonic = kwargs.get('require_monotonic_cost_decrease', True)
    has_converged = _get_stopping_criteria(kwargs.get('stopping_criterion', 'abs_delta_f'))
    dictionary_solver_kwargs = kwargs.get('dictionary_solver_kwargs', {})
    weights_solver_kwargs = kwargs.get('weights_solver_kwargs', {})
    scale_factors_solver_kwargs = kwargs.get('scale_factors_solver_kwargs', {})

    alpha = np.asarray(alpha, dtype=np.float64)
    ctx.set_state(dictionary, weights, alpha)
    new_cost = ctx.prepare()

    iter_times = []
    cost_deltas = []
    if verbose:
        print("*** {}: n_components = {:d} ***".format(label, n_components))
        print('{:<12s} | {:<13s} | {:<13s} | {:<12s}'.format(
            'Iteration', 'Cost', 'Cost delta', 'Time'))
        print(80 * '-')

    if _DEVICE_LOOP:
        # the loop runs on the device, monotonicity check and stopping rule included (and, for
        # delta != 0, the k-vector scale-factor SPG of reference :243-258 as one small kernel);
        # the host reads one status record per batch of iterations (aa_iterate).  verbose: one
        # batch per call so the table can be printed.
        scale_on = bool(update_scale_factors and delta != 0)
        stop_name = kwargs.get('stopping_criterion', 'abs_delta_f')
        # float32 data (this build's throughput mode): the trace-form cost cancels tr(XX')/n
        # down to the residual, so it carries ~eps_float32 * tr(XX')/n of noise per evaluation;
        # an "increase" inside that band is not one (archetypal_analysis.py:167-174 compares
        # float64 costs, where the band is below every sensible tolerance)
        mono_tol = tolerance
        if ctx.dtype_code == _backend.AA_F32:
            mono_tol = max(tolerance, 8 * 6e-8 * ctx.data_trace() / weights.shape[0])
        n_iter = -1
        done = False
        while not done and n_iter + 1 < max_iterations:
            budget = max_iterations - (n_iter + 1)
            chunk = min(budget, _DEVICE_LOOP_BATCH) if verbose else budget
            start_time = time.perf_counter()
            costs, st = ctx.iterate(new_cost, chunk, tolerance, stop_name, require_monotonic,
                                    update_dictionary, update_weights, dictionary_solver_kwargs,
                                    weights_solver_kwargs, check_every=_DEVICE_LOOP_BATCH,
                                    mono_tolerance=mono_tol, delta=delta if scale_on else 0.0,
                                    scale_kw=scale_factors_solver_kwargs if scale_on else None)
            elapsed = time.perf_counter() - start_time
            ran = max(st.reserved, 1)
            per_iter = elapsed / ran
            if update_dictionary:
                _warn_from_spg_flags(st)
            if st.error_stage:
                raise RuntimeError('factorization cost increased after {} update'.format(
                    {1: 'dictionary', 2: 'weights', 3: 'scale factors'}[st.error_stage]))
            starts = np.concatenate(([new_cost], costs[1::2][:-1]))
            finals = costs[1::2]
            for j in range(st.n_iter + 1):
                iter_times.append(per_iter)
                cost_deltas.append(finals[j] - starts[j])
                if verbose:
                    print('{:12d} | {: 12.6e} | {: 12.6e} | {: 12.6e}'.format(
                        n_iter + 2 + j, finals[j], finals[j] - starts[j], per_iter))
            n_iter += st.n_iter + 1
            new_cost = st.cost
            if st.converged:
                if verbose:
                    print('*** Converged at iteration {:d} ***'.format(n_iter + 1))
                done = True
        dictionary, weights, alpha = ctx.get_state()
        return (weights, dictionary, alpha, new_cost, n_iter, np.mean(iter_times), cost_deltas)

    # the same loop driven from the host one update at a time (scale factors with the generic
    # host spg): kept as the cross-check of the device loop (tests set _DEVICE_LOOP = False)
    n_iter = -1
    for n_iter in range(max_iterations):
        start_time = time.perf_counter()
        old_cost = new_cost

        if update_scale_factors and delta != 0:
            ZtZ, CKCt, CKZ, trace = ctx.grams()
            alpha = _update_kernel_aa_scale_factors(alpha, trace, CKZ, ZtZ, CKCt, delta,
                                                    **scale_factors_solver_kwargs)
            new_cost = _scaled_cost(ctx, alpha)
            _check_if_cost_increased(old_cost, new_cost, tolerance, 'scale factors',
                                     require_decrease=require_monotonic)

        if update_dictionary:
            stats = ctx.dictionary_update(**dictionary_solver_kwargs)
            _warn_from_spg_flags(stats)
            new_cost = ctx.cost()
            _check_if_cost_increased(old_cost, new_cost, tolerance, 'dictionary',
                                     require_decrease=require_monotonic)

        if update_weights:
            ctx.weights_update(**weights_solver_kwargs)
            new_cost = ctx.cost()
            _check_if_cost_increased(old_cost, new_cost, tolerance, 'weights',
                                     require_decrease=require_monotonic)

        end_time = time.perf_counter()
        iter_times.append(end_time - start_time)
        cost_deltas.append(new_cost - old_cost)
        if verbose:
            print('{:12d} | {: 12.6e} | {: 12.6e} | {: 12.6e}'.format(
                n_iter + 1, new_cost, new_cost - old_cost, end_time - start_time))
        if has_converged(old_cost, new_cost, tolerance):
            if verbose:
                print('*** Converged at iteration {:d} ***'.format(n_iter + 1))
            break

    dictionary, weights, _ = ctx.get_state()
    return (weights, dictionary, alpha, new_cost, n_iter, np.mean(iter_times), cost_deltas)


def _iterate_kernel_aa(K, weights, dictionary, alpha, delta=0,
                       update_weights=True, update_dictionary=True,
                       update_scale_factors=True, tolerance=1e-6,
                       max_iterations=1000, verbose=0, **kwargs):
    """Iterate kernel AA until convergence (reference :399-531); returns
    ``(weights, dictionary, alpha, cost, n_iter, mean_iter_time, cost_deltas)``."""
    with _backend.Context(dtype=np.float64) as ctx:
        ctx.set_data(np.asarray(K, dtype=np.float64), form=_backend.FORM_KERNEL)
        return _iterate_on_device(ctx, "Kernel AA", weights, dictionary, alpha, delta,
                                  update_weights, update_dictionary, update_scale_factors,
                                  tolerance, max_iterations, verbose, **kwargs)


def _iterate_aa(X, weights, dictionary, alpha, delta=0,
                update_weights=True, update_dictionary=True,
                update_scale_factors=True, tolerance=1e-6,
                max_iterations=1000, verbose=0, **kwargs):
    """Iterate AA until convergence (reference :534-670); same 7-tuple.  ``dtype`` may
    be passed in kwargs ('float64' default = reference arithmetic; 'float32' = MFMA
    throughput mode)."""
    with _backend.Context(dtype=kwargs.pop('dtype', None)) as ctx:
        ctx.set_data(X, form=_backend.FORM_DATA)
        return _iterate_on_device(ctx, "AA", weights, dictionary, alpha, delta,
                                  update_weights, update_dictionary, update_scale_factors,
                                  tolerance, max_iterations, verbose, **kwargs)


def _fit_on_data_matrix(self, data, linear_kernel, label, dictionary=None, weights=None, alpha=None,
                        update_dictionary=True, update_weights=True, update_scale_factors=True,
                        **kwargs):
    """Body of ``ArchetypalAnalysis._aa`` (reference :1036-1118) and of ``KernelAA`` on the
    implicit linear kernel (``linear_kernel``: the data matrix stands in for K = X X')."""
    on_device = isinstance(data, DeviceData)      # preprocessed on the GPU (preprocessing.py)
    if not on_device:
        data = np.asarray(data)
    n_samples = data.shape[0]
    if self.n_components is None:
        self.n_components = data.shape[1]
        self._n_components_defaulted_from = 'n_features'
    self._check_hyper_parameters()
    shape_only = _ShapeOnly(n_samples)
    if (kwargs.get('_draw_only', False) and self.init == 'random' and not on_device and dictionary is None
            and weights is None and alpha is None and update_dictionary and update_weights):
        # restarts.fit_restarts: the start factors of a random start need no device (and no checksum of
        # the data matrix per restart)
        return self._resolve_factors(
            n_samples, None, None, None, True, True,
            lambda: _initialize_kernel_aa_dictionary(shape_only, self.n_components, init='random',
                                                     random_state=self.random_state),
            lambda: _initialize_kernel_aa_weights(shape_only, self.n_components, init='random',
                                                  random_state=self.random_state))

    # the data matrix stays on the device between fits of the same array (the drivers' n_init
    # restarts, bin/run_hadisst_aa.py:158-172): only the start factors travel
    # one process per GPU (torch.distributed.run) with CONVEX_DIM_RED_DISTRIBUTED=1: this rank keeps
    # its row block of the data, RCCL all-reduces the Gram products, every rank gets the full factors
    distributed = _backend.distributed_env() is not None and not on_device
    draw_ctx = kwargs.get('_draw_ctx') if kwargs.get('_draw_only', False) else None
    with (_backend.borrowed(draw_ctx) if draw_ctx is not None else      # fit_restarts: one context for all draws
          data.borrow() if on_device else
          _backend.sharded_context(data, form=_backend.FORM_DATA, dtype=self.dtype) if distributed else
          _backend.resident_context(data, form=_backend.FORM_DATA, dtype=self.dtype)) as ctx:

        def init_dictionary():
            init = 'furthest_sum' if self.init is None else self.init
            if init == 'furthest_sum':
                rng = self.random_state
                start_index = kwargs.get('start_index', None)
                if start_index is None:
                    start_index = rng.randint(n_samples)
                exclude = kwargs.get('exclude', None)
                if exclude is None:
                    exclude = np.array([], dtype='i8')
                columns = kwargs['_cache'].setdefault('distance_columns', {}) if '_cache' in kwargs else None
                selected = _furthest_sum_on_device(
                    ctx, n_samples, self.n_components, start_index,
                    kwargs.get('n_extra_steps', 10), exclude, cache=columns)
                return _one_hot_rows(selected, n_samples, np.float64)
            return _initialize_kernel_aa_dictionary(shape_only, self.n_components, init=init,
                                                    random_state=self.random_state)

        dictionary, weights, alpha = self._resolve_factors(
            n_samples, dictionary, weights, alpha, update_dictionary, update_weights,
            init_dictionary,
            lambda: _initialize_kernel_aa_weights(shape_only, self.n_components,
                                                  init=self.init,
                                                  random_state=self.random_state))
        if kwargs.get('_draw_only', False):      # restarts.fit_restarts: start factors only
            return dictionary, weights, alpha
        self.weights = weights.copy()
        self.dictionary = dictionary.copy()
        self.alpha = alpha.copy()

        ctx.set_linear_kernel(linear_kernel)         # the context may be a reused one
        result = _iterate_on_device(
            ctx, label, self.weights, self.dictionary, self.alpha, self.delta,
            update_weights, update_dictionary, update_scale_factors, self.tolerance,
            self.max_iterations, self.verbose, **self._solver_kwargs())
        out = self._finish(result)
        self._cx = ctx.archetypes()
    return out


# ----------------------------------------------------------------------------
# estimators
# ----------------------------------------------------------------------------
class _BaseAA(object):
    _whom = '_kernel_aa'

    def __init__(self, n_components, delta=0, init=None,
                 tolerance=1e-6, max_iterations=1000, verbose=0,
                 random_state=None, **kwargs):
        self.n_components = n_components
        self.delta = delta
        self.init = init
        self.tolerance = tolerance
        self.max_iterations = max_iterations
        self.verbose = verbose
        self.random_state = check_random_state(random_state)
        self.require_monotonic_cost_decrease = kwargs.get('require_monotonic_cost_decrease', True)
        self.stopping_criterion = kwargs.get('stopping_criterion', 'abs_delta_f')
        self.dtype = kwargs.get('dtype', None)          # MI355X build: 'float64' | 'float32'

        self.weights = None
        self.dictionary = None
        self.alpha = None
        self.cost = 0
        self.n_iter = 0
        self.avg_time_per_iter = 0
        self.cost_deltas = None

        self.weights_solver_kwargs = kwargs.get('weights_solver_kwargs', {})
        self.dictionary_solver_kwargs = kwargs.get('dictionary_solver_kwargs', {})
        self.scale_factors_solver_kwargs = kwargs.get('scale_factors_solver_kwargs', {})

    def _check_hyper_parameters(self):
        if not isinstance(self.n_components, INTEGER_TYPES) or self.n_components <= 0:
            raise ValueError('Number of components must be a positive integer;'
                             ' got (n_components=%r)' % self.n_components)
        _backend.check_component_count(self.n_components, self._whom,
                                       getattr(self, '_n_components_defaulted_from', None))
        if not isinstance(self.max_iterations, INTEGER_TYPES) or self.max_iterations <= 0:
            raise ValueError('Maximum number of iterations must be a positive '
                             'integer; got (max_iterations=%r)' % self.max_iterations)
        if not isinstance(self.tolerance, numbers.Number) or self.tolerance < 0:
            raise ValueError('Tolerance for stopping criteria must be '
                             'positive; got (tolerance=%r)' % self.tolerance)

    def _resolve_factors(self, n_samples, dictionary, weights, alpha, update_dictionary,
                         update_weights, init_dictionary, init_weights):
        """Initialisation dispatch of the reference (:799-830, :1047-1078); the two
        callables produce a dictionary / weights from ``self.random_state``."""
        k = self.n_components
        whom = self._whom
        if self.init == 'custom':
            _check_init_weights(weights, (n_samples, k), '%s (input weights)' % whom)
            _check_init_dictionary(dictionary, (k, n_samples), '%s (input dictionary)' % whom)
            _check_init_scale_factors(alpha, self.delta, (k,), '%s (input scale factors)' % whom)
        elif not update_dictionary and update_weights:
            _check_init_dictionary(dictionary, (k, n_samples), '%s (input dictionary)' % whom)
            weights = init_weights()
        elif update_dictionary and not update_weights:
            _check_init_weights(weights, (n_samples, k), '%s (input weights)' % whom)
            dictionary = init_dictionary()
        else:
            dictionary = init_dictionary()
            weights = init_weights()
        if alpha is None:
            alpha = _initialize_kernel_aa_scale_factors_random(
                k, delta=self.delta, random_state=self.random_state)
        else:
            _check_init_scale_factors(alpha, self.delta, (k,), '%s (input scale factors)' % whom)
        return dictionary, weights, alpha

    def _solver_kwargs(self):
        return dict(require_monotonic_cost_decrease=self.require_monotonic_cost_decrease,
                    stopping_criterion=self.stopping_criterion,
                    weights_solver_kwargs=self.weights_solver_kwargs,
                    dictionary_solver_kwargs=self.dictionary_solver_kwargs,
                    scale_factors_solver_kwargs=self.scale_factors_solver_kwargs)

    def _finish(self, result):
        (self.weights, self.dictionary, self.alpha, cost, n_iter, avg_time, cost_deltas) = result
        if n_iter == self.max_iterations and self.tolerance > 0:
            warnings.warn('Maximum number of iterations %d reached.' % self.max_iterations,
                          UserWarning)
        return cost, n_iter, avg_time, cost_deltas


class KernelAA(_BaseAA):
    """Kernel archetypal analysis on a given kernel matrix (reference :673-910).

    Parameters and attributes are those of the reference class: ``n_components``,
    ``delta``, ``init`` (None | 'random' | 'furthest_sum' | 'custom'), ``tolerance``,
    ``max_iterations``, ``verbose``, ``random_state`` and the keyword dictionaries
    ``weights_solver_kwargs``, ``dictionary_solver_kwargs``,
    ``scale_factors_solver_kwargs``, ``require_monotonic_cost_decrease``,
    ``stopping_criterion``; after fitting: ``weights``, ``dictionary``, ``alpha``,
    ``cost``, ``n_iter``, ``avg_time_per_iter``, ``cost_deltas``.
    """
    _whom = '_kernel_aa'

    def _kernel_aa(self, kernel, dictionary=None, weights=None, alpha=None,
                   update_dictionary=True, update_weights=True,
                   update_scale_factors=True, **kwargs):
        features = kwargs.pop('features', False)
        which = kwargs.pop('implicit_kernel', 'linear')
        gamma = kwargs.pop('gamma', None)
        if features and which == 'rbf':
            return self._kernel_aa_rbf(kernel, gamma, dictionary, weights, alpha, update_dictionary,
                                       update_weights, update_scale_factors, **kwargs)
        if which not in ('linear', 'rbf') or (which == 'rbf' and not features):
            raise ValueError("kernel must be 'linear' or 'rbf' (with features=True); got %r" % (which,))
        if features:
            # `kernel` is the n x p feature matrix X (or a DeviceData) of the linear
            # kernel K = X X', which is never formed -- n = 100 000 would be 80 GB.  Same
            # algorithm, conventions and RNG order as the explicit-kernel path below; FurthestSum
            # runs on ||x_i - x_j||^2 = K_ii - 2 K_ij + K_jj computed from X.
            if self.n_components is None:
                self.n_components = kernel.shape[0]
                self._n_components_defaulted_from = 'n_samples'
            out = _fit_on_data_matrix(self, kernel, True, "Kernel AA", dictionary, weights, alpha,
                                      update_dictionary, update_weights, update_scale_factors,
                                      **kwargs)
            self.__dict__.pop('_cx', None)
            return out
        kernel = np.asarray(kernel)
        n_samples = kernel.shape[0]
        if kernel.shape[1] != n_samples:
            raise ValueError('Expected square kernel matrix in %s. '
                             'Got shape %s' % ('kernel_aa', kernel.shape))
        if self.n_components is None:
            self.n_components = n_samples
            self._n_components_defaulted_from = 'n_samples'
        self._check_hyper_parameters()

        dictionary, weights, alpha = self._resolve_factors(
            n_samples, dictionary, weights, alpha, update_dictionary, update_weights,
            lambda: _initialize_kernel_aa_dictionary(kernel, self.n_components, init=self.init,
                                                     random_state=self.random_state, **kwargs),
            lambda: _initialize_kernel_aa_weights(kernel, self.n_components, init=self.init,
                                                  random_state=self.random_state))
        self.weights = weights.copy()
        self.dictionary = dictionary.copy()
        self.alpha = alpha.copy()

        return self._finish(_iterate_kernel_aa(
            kernel, self.weights, self.dictionary, self.alpha, delta=self.delta,
            update_weights=update_weights, update_dictionary=update_dictionary,
            update_scale_factors=update_scale_factors, tolerance=self.tolerance,
            max_iterations=self.max_iterations, verbose=self.verbose, **self._solver_kwargs()))

    def _kernel_aa_rbf(self, X, gamma, dictionary, weights, alpha, update_dictionary, update_weights,
                       update_scale_factors, **kwargs):
        """SURVEY 8(f4), the alternative it names: the RBF kernel ``exp(-gamma ||x_i - x_j||^2)`` of the
        rows of ``X``, never formed -- the kernel form of the algorithm (reference :399-531, :673-910)
        with every product ``C K`` / ``K Z`` computed as one fused distance + exp + multiply pass over
        the features (aa_set_rbf_features).  Same conventions, initialisers and RNG order as the
        explicit-kernel path: ``KernelAA(...).fit_transform(rbf_kernel(X, gamma=g))`` is what it
        reproduces (tests/test_gpu_parity.py::test_kernel_aa_on_the_implicit_rbf_kernel)."""
        X = np.asarray(X, dtype=np.float64)
        if X.ndim != 2:
            raise ValueError('Expected a feature matrix (n_samples x n_features); got shape %s' % (X.shape,))
        if gamma is None:
            gamma = 1.0 / X.shape[1]                      # scikit-learn's rbf_kernel default
        n_samples = X.shape[0]
        if self.n_components is None:
            self.n_components = n_samples
            self._n_components_defaulted_from = 'n_samples'
        self._check_hyper_parameters()
        shape_only = _ShapeOnly(n_samples)
        with _backend.Context(dtype=np.float64) as ctx:
            ctx.set_rbf_features(X, gamma)

            def init_dictionary():
                init = 'furthest_sum' if self.init is None else self.init
                if init == 'furthest_sum':
                    rng = self.random_state
                    start_index = kwargs.get('start_index', None)
                    if start_index is None:
                        start_index = rng.randint(n_samples)
                    exclude = kwargs.get('exclude', None)
                    if exclude is None:
                        exclude = np.array([], dtype='i8')
                    selected = _furthest_sum_on_device(ctx, n_samples, self.n_components, start_index,
                                                       kwargs.get('n_extra_steps', 10), exclude)
                    return _one_hot_rows(selected, n_samples, np.float64)
                return _initialize_kernel_aa_dictionary(shape_only, self.n_components, init=init,
                                                        random_state=self.random_state)

            dictionary, weights, alpha = self._resolve_factors(
                n_samples, dictionary, weights, alpha, update_dictionary, update_weights, init_dictionary,
                lambda: _initialize_kernel_aa_weights(shape_only, self.n_components, init=self.init,
                                                      random_state=self.random_state))
            self.weights = weights.copy()
            self.dictionary = dictionary.copy()
            self.alpha = alpha.copy()
            return self._finish(_iterate_on_device(
                ctx, "Kernel AA", self.weights, self.dictionary, self.alpha, self.delta, update_weights,
                update_dictionary, update_scale_factors, self.tolerance, self.max_iterations, self.verbose,
                **self._solver_kwargs()))

    def fit_transform(self, data, dictionary=None, weights=None, alpha=None, **kwargs):
        """Factorise the kernel matrix ``data`` (n x n) and return the weights.  With
        ``features=True`` (an extension of the reference's signature) ``data`` is the n x p
        feature matrix of the linear kernel ``data.dot(data.T)`` -- or, with ``kernel='rbf',
        gamma=g``, of the RBF kernel ``exp(-g ||x_i - x_j||^2)`` -- which is then never formed."""
        if 'kernel' in kwargs:                 # (`kernel` is also the name of _kernel_aa's first argument)
            kwargs['implicit_kernel'] = kwargs.pop('kernel')
        self.cost, self.n_iter, self.avg_time_per_iter, self.cost_deltas = self._kernel_aa(
            data, dictionary=dictionary, weights=weights, alpha=alpha, **kwargs)
        return self.weights

    def fit(self, kernel, **kwargs):
        self.fit_transform(kernel, **kwargs)
        return self


class ArchetypalAnalysis(_BaseAA):
    """Standard archetypal analysis ``min ||X - Z C X||_F^2`` (reference :913-1215).

    Same constructor, attributes (plus ``archetypes``) and methods as the reference:
    ``fit_transform(X)``, ``transform(X) -> (weights, cost)``, ``inverse_transform``.
    As in the reference there is no ``fit`` method.  Extra keyword: ``dtype``
    ('float64' default = reference arithmetic, 'float32' = fp32-MFMA throughput mode).
    """
    _whom = '_aa'

    def __init__(self, n_components, delta=0, init=None, tolerance=1e-6, max_iterations=1000,
                 verbose=0, random_state=None, **kwargs):
        super(ArchetypalAnalysis, self).__init__(
            n_components, delta=delta, init=init, tolerance=tolerance,
            max_iterations=max_iterations, verbose=verbose, random_state=random_state, **kwargs)
        self.archetypes = None

    def _aa(self, data, dictionary=None, weights=None, alpha=None,
            update_dictionary=True, update_weights=True, update_scale_factors=True, **kwargs):
        return _fit_on_data_matrix(self, data, False, "AA", dictionary, weights, alpha,
                                   update_dictionary, update_weights, update_scale_factors, **kwargs)

    def fit_transform(self, data, dictionary=None, weights=None, alpha=None, **kwargs):
        """Factorise ``data`` (n_samples x n_features) and return the weights."""
        self.cost, n_iter_, avg_time_, cost_deltas_ = self._aa(
            data, dictionary=dictionary, weights=weights, alpha=alpha, **kwargs)
        cx = self.__dict__.pop('_cx')
        if self.delta != 0:
            self.dictionary = np.dot(np.diag(self.alpha), self.dictionary)
            cx = self.alpha[:, np.newaxis] * cx
        self.archetypes = cx                     # = dictionary.dot(data), reference :1144
        self.n_iter = n_iter_
        self.avg_time_per_iter = avg_time_
        self.cost_deltas = cost_deltas_
        return self.weights

    def transform(self, data):
        """Weights of new samples for the fitted archetypes, and their cost (reference
        :1151-1199): per-sample QPs with A = (CX)(CX)', b_t = -(CX) x_t from fresh random
        starting weights, ``max_iterations`` SPG passes at most, then 0.5 ||X - W (CX)||_F^2 / n.
        One context, one upload: X (CX)' is the row-local GEMM, the QPs and the residual norm run
        on the resident data (csrc/solver.hip: aa_gpnh_set_factors / _weights_update /
        _residual_cost, which are the "data times a k x p dictionary" entry points)."""
        on_device = isinstance(data, DeviceData)
        if not on_device:
            data = np.asarray(data)
            if data.dtype != np.float32:
                data = np.asarray(data, dtype=np.float64)
        n_samples = data.shape[0]
        kw = dict(self.weights_solver_kwargs)
        kw['max_iterations'] = self.max_iterations            # reference :1194
        archetypes = np.asarray(self.archetypes, dtype=np.float64)
        CKCt = archetypes.dot(archetypes.T)                   # k x k
        initial_weights = right_stochastic_matrix((n_samples, self.n_components),
                                                  random_state=self.random_state)
        if on_device:
            manager = data.borrow()
        else:
            manager = _backend.Context(dtype=np.float64 if data.dtype == np.float64 else self.dtype)
        with manager as ctx:
            if not on_device:
                ctx.set_data(data, form=_backend.FORM_DATA)
            ctx.gpnh_set_factors(self.n_components, W=archetypes.T, Z=initial_weights)   # X (CX)'
            ctx.gpnh_weights_update(CKCt, **kw)
            self.weights = ctx.gpnh_get_weights()
            cost = ctx.gpnh_residual_cost()
        return self.weights, cost

    def inverse_transform(self, weights):
        return weights.dot(self.archetypes)
