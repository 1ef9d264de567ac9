"""GPNH-regularised convex coding on MI355X.

Mirrors reference src/convex_dim_red/gpnh_convex_coding.py (class ``GPNHConvexCoding``
and the test-visible ``_gpnh_cost``, ``_iterate_gpnh_convex_coding``,
``_update_gpnh_dictionary``, ``_update_gpnh_weights``).

The whole alternating loop runs on the device (csrc/solver.hip: aa_gpnh_iterate): Z'X
(reduce-over-rows GEMM), the k x k regularised normal equations for the dictionary
(reference :213-226; Cholesky in LDS), X W (row-local GEMM), W'W, the GPNH penalty, the
cost, the n per-sample simplex QPs, the monotonicity check and the stopping rule; the host
reads one status record per batch of iterations.  When the normal equations are not
positive definite (an unused component) the host loop with ``numpy.linalg.lstsq`` -- the
reference's solver -- takes over.
"""
from __future__ import absolute_import, division, print_function

import numbers
import time
import warnings

import numpy as np
from sklearn.utils import check_array, check_random_state

from . import _backend
from .furthest_sum import furthest_sum_from_columns
from .preprocessing import DeviceData
from .stochastic_matrices import right_stochastic_matrix
from .validation_utils import check_unit_axis_sums, check_array_shape

INTEGER_TYPES = (numbers.Integral, np.integer)
INITIALIZATION_METHODS = (None, 'random', 'furthest_sum',)


def _check_init_weights(weights, shape, whom):
    weights = check_array(weights)
    check_array_shape(weights, shape, whom)
    check_unit_axis_sums(weights, whom, axis=1)


def _check_init_dictionary(dictionary, shape, whom):
    check_array_shape(check_array(dictionary), shape, whom)


def _check_if_cost_increased(old, new, tolerance, stage, require_decrease=True):
    if (new > old) and (abs(new - old) > tolerance) and require_decrease:
        raise RuntimeError('factorization cost increased after {} update'.format(stage))


def _get_stopping_criteria(stopping_criterion):
    if stopping_criterion == 'abs_delta_f':
        return lambda old_cost, new_cost, tolerance: abs(new_cost - old_cost) < tolerance
    if stopping_criterion == 'rel_delta_f':
        return lambda old_cost, new_cost, tolerance: (
            abs((new_cost - old_cost) / max(abs(new_cost), abs(old_cost))) < tolerance)
    raise ValueError("unsupported stopping criterion '%s'" % stopping_criterion)


def _gpnh_regularization(dictionary):
    """GPNH penalty: mean squared pairwise distance between dictionary columns
    (reference :179-196), via the Gram matrix of the p x k dictionary."""
    n_features, n_components = dictionary.shape
    if n_components == 1:
        return 0.0
    gram = dictionary.T.dot(dictionary)
    sq = np.diag(gram)
    pair = sq[:, np.newaxis] + sq[np.newaxis, :] - 2.0 * gram
    total = np.triu(pair, 1).sum()
    return 2.0 / (n_components * n_features * (n_components - 1.0)) * total


def _gpnh_matrix(n_features, n_components):
    """GW of reference :296-300."""
    if n_components > 1:
        return (4.0 / (n_features * n_components * (n_components - 1))) * (
            n_components * np.eye(n_components) - 1)
    return np.zeros((n_components, n_components))


def _gpnh_cost(data, weights, dictionary, lambda_W=0):
    """0.5 ||X - Z W'||_F^2 / n + lambda_W * penalty (reference :199-210); the residual
    norm is evaluated on the device in residual form."""
    data = np.asarray(data, dtype=np.float64)
    dictionary = np.asarray(dictionary, dtype=np.float64)
    with _backend.Context(dtype=np.float64) as ctx:
        ctx.set_data(data)
        ctx.gpnh_set_factors(dictionary.shape[1], W=dictionary, Z=weights)
        cost = ctx.gpnh_residual_cost()
    if lambda_W != 0:
        cost += lambda_W * _gpnh_regularization(dictionary)
    return cost


def _solve_dictionary(ZtX, ZtZ, GW, n_samples, lambda_W):
    lhs = ZtZ / n_samples + lambda_W * GW
    rhs = ZtX / n_samples
    return np.linalg.lstsq(lhs, rhs, rcond=None)[0].T


def _update_gpnh_dictionary(X, weights, ZtZ, GW, lambda_W=0):
    """Regularised least-squares dictionary update (reference :213-226)."""
    X = np.asarray(X, dtype=np.float64)
    with _backend.Context(dtype=np.float64) as ctx:
        ctx.set_data(X)
        ctx.gpnh_set_factors(weights.shape[1], Z=weights)
        ZtX = ctx.gpnh_reduce(want_ztx=True, want_trace=False)[0]
    return _solve_dictionary(ZtX, ZtZ, GW, X.shape[0], lambda_W)


def _update_gpnh_weights(X, weights, dictionary, **solver_kwargs):
    """Per-sample simplex QPs with A = W'W, b_t = -(X W)[t] (reference :254-279)."""
    X = np.asarray(X, dtype=np.float64)
    dictionary = np.asarray(dictionary, dtype=np.float64)
    with _backend.Context(dtype=np.float64) as ctx:
        ctx.set_data(X)
        ctx.gpnh_set_factors(dictionary.shape[1], W=dictionary, Z=weights)
        ctx.gpnh_weights_update(dictionary.T.dot(dictionary), **solver_kwargs)
        return ctx.gpnh_get_weights()


def _iterate_on_device(ctx, n_features, weights, dictionary, lambda_W, update_weights,
                       update_dictionary, tolerance, max_iterations, verbose, **kwargs):
    n_samples, n_components = weights.shape
    require_monotonic = kwargs.get('require_monotonic_cost_decrease', True)
    has_converged = _get_stopping_criteria(kwargs.get('stopping_criterion', 'abs_delta_f'))
    dictionary_solver_kwargs = kwargs.get('dictionary_solver_kwargs', {})
    weights_solver_kwargs = kwargs.get('weights_solver_kwargs', {})
    if update_dictionary and dictionary_solver_kwargs:
        # the reference forwards these to a function that accepts none (:348-350)
        raise TypeError("_update_gpnh_dictionary() got an unexpected keyword argument %r"
                        % sorted(dictionary_solver_kwargs)[0])

    dictionary = np.asarray(dictionary, dtype=np.float64)
    ctx.gpnh_set_factors(n_components, W=dictionary, Z=weights)
    if _DEVICE_LOOP:
        out = _device_loop(ctx, weights.shape[0], lambda_W, update_weights, update_dictionary,
                           tolerance, max_iterations, verbose, require_monotonic,
                           kwargs.get('stopping_criterion', 'abs_delta_f'), weights_solver_kwargs)
        if out is not None:
            return out
        ctx.gpnh_set_factors(n_components, W=dictionary, Z=weights)      # singular system: host path
    return _host_loop(ctx, n_features, weights, dictionary, lambda_W, update_weights,
                      update_dictionary, tolerance, max_iterations, verbose, require_monotonic,
                      has_converged, weights_solver_kwargs)


_DEVICE_LOOP = True
_DEVICE_LOOP_BATCH = 8          # outer iterations between two host polls of the device loop


def _device_loop(ctx, n_samples, lambda_W, update_weights, update_dictionary, tolerance,
                 max_iterations, verbose, require_monotonic, stop_name, weights_solver_kwargs):
    """The loop of reference :334-399 on the device (csrc/solver.hip: aa_gpnh_iterate): Z'X, the
    k x k regularised normal equations (Cholesky), X W, W'W, the penalty, the cost, the QPs, the
    monotonicity check and the stopping rule; the host reads one status record per batch of
    iterations.  Returns None when the normal equations turn out not to be positive definite
    (an unused component): the caller then runs the host loop with numpy.linalg.lstsq, the
    reference's solver."""
    mono_tol = tolerance
    if ctx.dtype_code == _backend.AA_F32:
        mono_tol = max(tolerance, 8 * 6e-8 * ctx.data_trace() / n_samples)
    iter_times, cost_deltas = [], []
    if verbose:
        print("*** GPNH convex coding: n_components = {:d} ***".format(ctx.k))
        print('{:<12s} | {:<13s} | {:<13s} | {:<12s}'.format('Iteration', 'Cost', 'Cost delta', 'Time'))
        print(100 * '-')
    n_iter, new_cost, done = -1, None, False
    while not done and n_iter + 1 < max_iterations:
        budget = max_iterations - (n_iter + 1)
        chunk = min(budget, _DEVICE_LOOP_BATCH) if verbose else budget
        start_time = time.perf_counter()
        cost0, costs, st = ctx.gpnh_iterate(lambda_W, chunk, tolerance, stop_name, require_monotonic,
                                            update_dictionary, update_weights, weights_solver_kwargs,
                                            check_every=_DEVICE_LOOP_BATCH, mono_tolerance=mono_tol)
        elapsed = time.perf_counter() - start_time
        if st.error_stage == 3:
            return None
        if st.error_stage:
            raise RuntimeError('factorization cost increased after {} update'.format(
                'dictionary' if st.error_stage == 1 else 'weights'))
        per_iter = elapsed / max(st.reserved, 1)
        starts = np.concatenate(([cost0], costs[1::2][:-1]))
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
    weights = ctx.gpnh_get_weights()
    dictionary = ctx.gpnh_get_dictionary()
    return (weights, dictionary, new_cost, n_iter, np.mean(iter_times), cost_deltas)


def _host_loop(ctx, n_features, weights, dictionary, lambda_W, update_weights, update_dictionary,
               tolerance, max_iterations, verbose, require_monotonic, has_converged,
               weights_solver_kwargs):
    """One update at a time from the host, the dictionary solved with numpy.linalg.lstsq exactly
    as the reference does (:213-226): the fallback for rank-deficient normal equations."""
    n_samples, n_components = weights.shape
    GW = _gpnh_matrix(n_features, n_components)
    trace_XtX = ctx.data_trace()
    _, ZtZ, trace_WtXtZ = ctx.gpnh_reduce(want_ztx=False)
    WtW = dictionary.T.dot(dictionary)
    penalty = lambda_W * _gpnh_regularization(dictionary) if lambda_W != 0 else 0

    def current_cost():
        return (0.5 * (trace_XtX - 2 * trace_WtXtZ + ZtZ.dot(WtW).trace()) / n_samples + penalty)

    new_cost = current_cost()
    iter_times = []
    cost_deltas = []
    if verbose:
        print("*** GPNH convex coding: n_components = {:d} ***".format(n_components))
        print('{:<12s} | {:<13s} | {:<13s} | {:<12s}'.format(
            'Iteration', 'Cost', 'Cost delta', 'Time'))
        print(100 * '-')

    n_iter = -1
    for n_iter in range(max_iterations):
        start_time = time.perf_counter()
        old_cost = new_cost

        if update_dictionary:
            ZtX = ctx.gpnh_reduce(want_ztx=True, want_trace=False)[0]
            dictionary = _solve_dictionary(ZtX, ZtZ, GW, n_samples, lambda_W)
            ctx.gpnh_set_factors(n_components, W=dictionary)          # also X W
            WtW = dictionary.T.dot(dictionary)
            trace_WtXtZ = ctx.gpnh_reduce(want_ztx=False)[2]
            penalty = lambda_W * _gpnh_regularization(dictionary) if lambda_W != 0 else 0
            new_cost = current_cost()
            _check_if_cost_increased(old_cost, new_cost, tolerance, 'dictionary',
                                     require_decrease=require_monotonic)

        if update_weights:
            ctx.gpnh_weights_update(WtW, **weights_solver_kwargs)
            _, ZtZ, trace_WtXtZ = ctx.gpnh_reduce(want_ztx=False)
            new_cost = current_cost()
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

    weights = ctx.gpnh_get_weights()
    return (weights, dictionary, new_cost, n_iter, np.mean(iter_times), cost_deltas)


def _iterate_gpnh_convex_coding(X, weights, dictionary, lambda_W=0,
                                update_weights=True, update_dictionary=True,
                                tolerance=1e-6, max_iterations=1000, verbose=0, **kwargs):
    """Alternating updates until convergence (reference :282-402); returns
    ``(weights, dictionary, cost, n_iter, mean_iter_time, cost_deltas)``."""
    X = np.asarray(X)
    with _backend.Context(dtype=kwargs.pop('dtype', None)) as ctx:
        ctx.set_data(X)
        return _iterate_on_device(ctx, X.shape[1], weights, dictionary, lambda_W, update_weights,
                                  update_dictionary, tolerance, max_iterations, verbose, **kwargs)


class GPNHConvexCoding(object):
    """Convex coding with GPNH regularisation (reference :405-668): same constructor
    arguments (``n_components``, ``lambda_W``, ``init`` in None | 'random' |
    'furthest_sum' | 'custom', ``tolerance``, ``max_iterations``, ``verbose``,
    ``random_state``, keyword dictionaries), attributes (``weights``, ``dictionary``
    (n_features x n_components), ``cost``, ``n_iter``, ``avg_time_per_iter``,
    ``cost_deltas``) and methods (``fit_transform``, ``fit``, ``transform``,
    ``inverse_transform``)."""

    def __init__(self, n_components, lambda_W=0, init=None, tolerance=1e-6, max_iterations=1000,
                 verbose=0, random_state=None, **kwargs):
        self.n_components = n_components
        self.lambda_W = lambda_W
        self.init = init
        self.tolerance = tolerance
        self.max_iterations = max_iterations
        self.verbose = verbose
        self.random_state = check_random_state(random_state)
        self.require_monotonic_cost_decrease = kwargs.get('require_monotonic_cost_decrease', True)
        self.stopping_criterion = kwargs.get('stopping_criterion', 'abs_delta_f')
        self.dtype = kwargs.get('dtype', None)

        self.weights = None
        self.dictionary = None
        self.cost = 0
        self.n_iter = 0
        self.avg_time_per_iter = 0
        self.cost_deltas = None

        self.weights_solver_kwargs = kwargs.get('weights_solver_kwargs', {})
        self.dictionary_solver_kwargs = kwargs.get('dictionary_solver_kwargs', {})

    def _initial_dictionary(self, ctx, data, kwargs):
        init = 'random' if self.init is None else self.init
        n_samples, n_features = data.shape
        rng = self.random_state
        if init == 'random':                                   # reference :41-49
            cache = kwargs.get('_cache')                       # restarts.fit_restarts: the same data every time
            if cache is not None and 'abs_mean' in cache:
                abs_mean = cache['abs_mean']
            else:
                abs_mean = np.abs(data).mean()
                if cache is not None:
                    cache['abs_mean'] = abs_mean
            avg = np.sqrt(abs_mean / self.n_components)
            return avg * rng.randn(n_features, self.n_components)
        if init == 'furthest_sum':                             # reference :52-81
            start_index = kwargs.get('start_index', None)
            if start_index is None:
                start_index = rng.randint(n_samples)
            exclude = kwargs.get('exclude', None)
            if exclude is None:
                exclude = np.array([], dtype='i8')
            cache = kwargs['_cache'].setdefault('distance_columns', {}) if '_cache' in kwargs else {}
            from .archetypal_analysis import _furthest_sum_on_device
            selected = _furthest_sum_on_device(ctx, n_samples, self.n_components, start_index,
                                               kwargs.get('n_extra_steps', 10), exclude, cache=cache)
            return np.ascontiguousarray(np.asarray(data, dtype=np.float64)[selected].T)
        raise ValueError('Invalid init parameter: got %r instead of one of %r'
                         % (init, INITIALIZATION_METHODS))

    def _initial_weights(self, n_samples):
        init = 'random' if self.init is None else self.init
        if init in ('furthest_sum', 'random'):
            return right_stochastic_matrix((n_samples, self.n_components),
                                           random_state=self.random_state)
        raise ValueError('Invalid init parameter: got %r instead of one of %r'
                         % (init, INITIALIZATION_METHODS))

    def _gpnh_convex_coding(self, data, dictionary=None, weights=None,
                            update_dictionary=True, update_weights=True, **kwargs):
        on_device = isinstance(data, DeviceData)      # preprocessed on the GPU (preprocessing.py)
        if not on_device:
            data = np.asarray(data)
        n_samples, n_features = data.shape
        defaulted = None
        if self.n_components is None:
            self.n_components = n_features
            defaulted = 'n_features'
        if not isinstance(self.n_components, INTEGER_TYPES) or self.n_components <= 0:
            raise ValueError('Number of components must be a positive integer;'
                             ' got (n_components=%r)' % self.n_components)
        _backend.check_component_count(self.n_components, 'GPNHConvexCoding', defaulted)
        if not isinstance(self.max_iterations, INTEGER_TYPES) or self.max_iterations <= 0:
            raise ValueError('Maximum number of iterations must be a positive '
                             'integer; got (max_iterations=%r)' % self.max_iterations)
        if not isinstance(self.tolerance, numbers.Number) or self.tolerance < 0:
            raise ValueError('Tolerance for stopping criteria must be '
                             'positive; got (tolerance=%r)' % self.tolerance)
        k = self.n_components
        whom = '_gpnh_convex_coding'
        if (kwargs.get('_draw_only', False) and not on_device and update_dictionary and update_weights
                and self.init in (None, 'random')):
            # restarts.fit_restarts: the start factors of a random start need no device
            return self._initial_dictionary(None, data, kwargs), self._initial_weights(n_samples)
        # data resident across the drivers' n_init restarts (bin/run_jra55_pca_gpnh.py:123-136)
        distributed = _backend.distributed_env() is not None and not on_device   # see _backend.distributed_env
        draw_ctx = kwargs.get('_draw_ctx') if kwargs.get('_draw_only', False) else None
        with (_backend.borrowed(draw_ctx) if draw_ctx is not None else  # fit_restarts: one context for all draws
              data.borrow() if on_device else
              _backend.sharded_context(data, dtype=self.dtype) if distributed else
              _backend.resident_context(data, dtype=self.dtype)) as ctx:
            ctx.set_linear_kernel(False)             # a reused context may come from KernelAA(features=True)
            if on_device:
                data = data.to_host() if self.init != 'custom' and update_dictionary else data
            if self.init == 'custom':
                _check_init_weights(weights, (n_samples, k), whom + ' (input weights)')
                _check_init_dictionary(dictionary, (n_features, k), whom + ' (input dictionary)')
            elif not update_dictionary and update_weights:
                _check_init_dictionary(dictionary, (n_features, k), whom + ' (input dictionary)')
                weights = self._initial_weights(n_samples)
            elif update_dictionary and not update_weights:
                _check_init_weights(weights, (n_samples, k), whom + ' (input weights)')
                dictionary = self._initial_dictionary(ctx, data, kwargs)
            else:
                dictionary = self._initial_dictionary(ctx, data, kwargs)
                weights = self._initial_weights(n_samples)

            if kwargs.get('_draw_only', False):      # restarts.fit_restarts: start factors only
                return dictionary, weights
            self.weights = weights.copy()
            self.dictionary = np.array(dictionary, dtype=np.float64)

            (self.weights, self.dictionary, cost, n_iter, avg_time_per_iter, cost_deltas) = \
                _iterate_on_device(
                    ctx, n_features, self.weights, self.dictionary, self.lambda_W, update_weights,
                    update_dictionary, self.tolerance, self.max_iterations, self.verbose,
                    require_monotonic_cost_decrease=self.require_monotonic_cost_decrease,
                    stopping_criterion=self.stopping_criterion,
                    weights_solver_kwargs=self.weights_solver_kwargs,
                    dictionary_solver_kwargs=self.dictionary_solver_kwargs)

        if n_iter == self.max_iterations and self.tolerance > 0:
            warnings.warn('Maximum number of iterations %d reached.' % self.max_iterations,
                          UserWarning)
        return cost, n_iter, avg_time_per_iter, cost_deltas

    def fit_transform(self, data, dictionary=None, weights=None, **kwargs):
        """Fit to ``data`` (n_samples x n_features) and return the weights."""
        self.cost, self.n_iter, self.avg_time_per_iter, self.cost_deltas = \
            self._gpnh_convex_coding(data, dictionary=dictionary, weights=weights, **kwargs)
        return self.weights

    def fit(self, data, **kwargs):
        self.fit_transform(data, **kwargs)
        return self

    def transform(self, data):
        """Weights of ``data`` for the fitted dictionary, and the cost (reference
        :623-652; the extra keywords the reference passes there are not consumed by its
        initialisers and are dropped)."""
        cost_ = self._gpnh_convex_coding(data=data, dictionary=self.dictionary,
                                         update_dictionary=False, update_weights=True)[0]
        return self.weights, cost_

    def inverse_transform(self, weights):
        return weights.dot(self.dictionary.T)
