"""The drivers' restart loops, several fits at a time.

``bin/run_hadisst_aa.py:149-174`` / ``bin/run_jra55_pca_gpnh.py:112-138`` fit ``n_init`` fresh
models one after the other, all drawing their starting factors from ONE shared ``RandomState``,
and keep the one with the lowest cost.  For the problems those drivers run (161 MB / 15 MB
matrices) one fit leaves most of an MI355X idle -- an outer iteration is a few dozen dependent
launches of a few microseconds each -- so ``fit_restarts`` draws the starting factors of the
restarts in the drivers' order (so every restart starts exactly where it would in the sequential
loop; a worker thread draws while the device already iterates on the first restarts: ``_RestartFeed``),
and

* GPNH models, and AA models with the drivers' settings (one SPG iteration per dictionary update,
  any delta, fewer than 65 536 samples, k <= 16): lays the restarts SIDE BY SIDE in the component slots of
  one set of device arrays, where they share every launch of an outer iteration; a restart that stops
  hands its slot to the next one (``_fit_gpnh_slots`` / aa_gpnh_slots_*, ``_fit_aa_slots`` /
  aa_slots_*): 3.4-6x the sequential loop's speed on the JRA-55- and HadISST-shaped problems with
  ``n_init = 100``;
* other AA settings: runs the fits on worker threads, each on its own device context; the contexts
  of a device share one resident copy of the data (useful with ``devices=[...]``: whole restarts
  dealt over GPUs; on ONE GPU several fits at a time are not faster).

Every model ends with the attributes the sequential loop gives it; results are identical, restart
by restart.
"""
from __future__ import absolute_import, division

import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _backend
from . import archetypal_analysis as _aa_module
from .archetypal_analysis import ArchetypalAnalysis
from .gpnh_convex_coding import GPNHConvexCoding
from .preprocessing import DeviceData


slots_profile = {}        # seconds spent loading / iterating / fetching in the last side-by-side run


def _print_tables(models, tables, title, rule):
    """verbose: the per-iteration tables of the sequential loops (archetypal_analysis.py:566-584,655-661;
    gpnh_convex_coding.py:336-340,391-397), one restart after the other in restart order -- printed
    when all restarts are done, since they ran side by side."""
    for m, tab in zip(models, tables):
        if tab is None:
            continue
        finals, begins, per_iter, converged = tab
        print(title.format(m.n_components))
        print('{:<12s} | {:<13s} | {:<13s} | {:<12s}'.format('Iteration', 'Cost', 'Cost delta', 'Time'))
        print(rule * '-')
        for j in range(len(finals)):
            print('{:12d} | {: 12.6e} | {: 12.6e} | {: 12.6e}'.format(j + 1, finals[j], finals[j] - begins[j], per_iter))
        if converged:
            print('*** Converged at iteration {:d} ***'.format(len(finals)))


_GPNH_KEYS = ("n_components", "lambda_W", "tolerance", "max_iterations", "stopping_criterion", "verbose",
              "require_monotonic_cost_decrease", "dtype", "weights_solver_kwargs", "dictionary_solver_kwargs")
_AA_KEYS = ("n_components", "delta", "tolerance", "max_iterations", "stopping_criterion", "verbose",
            "require_monotonic_cost_decrease", "dtype", "weights_solver_kwargs", "dictionary_solver_kwargs",
            "scale_factors_solver_kwargs")


def _same_settings(m, m0):
    """A later restart can sit beside the first one: same class, same hyper-parameters."""
    if type(m) is not type(m0):
        return False
    keys = _AA_KEYS if isinstance(m0, ArchetypalAnalysis) else _GPNH_KEYS
    return all(getattr(m, a) == getattr(m0, a) for a in keys)


def _slots_eligible(m0, n_models, data):
    """GPNH restarts that can share one set of device arrays (every restart with the first one's
    hyper-parameters: _same_settings; the others take the generic path), at least two
    restarts of k components in the 64 component slots of the tall arrays.  Mirrors the
    preconditions of aa_gpnh_slots_begin (csrc/solver.hip): QPs of more than four passes take the
    AA slots' four-lane / wave-per-sample launch, which exists for fewer than 65 536 samples."""
    if not isinstance(m0, GPNHConvexCoding) or n_models < 2:
        return False
    k = m0.n_components
    # (weights QPs of at most four SPG passes -- the drivers' setting is one -- run in the
    # lane-per-sample kernel, longer ones in the four-lane and wave-per-sample kernels: per slot what a
    # single fit runs, so every restart gets the bits it gets alone)
    passes = m0.weights_solver_kwargs.get("max_iterations", 1000)
    memory = m0.weights_solver_kwargs.get("memory", 1)
    return (isinstance(k, int) and 1 <= k <= 16 and not m0.dictionary_solver_kwargs and passes >= 1
            and (memory <= 8 if passes <= 4 else (memory <= 1 and data.shape[0] < 65536))
            and _backend.distributed_env() is None)


def _gpnh_slot_count(m0, n_samples, n_slots):
    """Restarts side by side: 64 // k in the 64 component slots -- except for float32 data from
    32 768 samples on, where a single fit (32 component slots) runs the row-local pass that sums
    32-column fp32 pieces in float64 (k_row_local_f32_dma) and 64 slots would take the
    register-staged kernel with plain fp32 chains (csrc/kernels_gemm.hip: row_local_variant): the
    restarts stay within 32 slots there so that every restart keeps the bits -- and the accumulation
    accuracy -- it has alone."""
    k = m0.n_components
    room = 64
    if _backend.dtype_code(m0.dtype) == _backend.AA_F32 and n_samples > 32768 - 128:
        room = 32
    cap = max(1, room // k)
    return cap if n_slots is None else max(1, min(int(n_slots), cap))


class _SelectionRecorder(object):
    """Stands in for the draw context of a FurthestSum start on the feed's thread: notes what the
    selection is asked for and answers with a placeholder.  The selection -- a chain of ~60 device
    launches, no random numbers -- runs when the restart takes its slot, on the slots' own context
    (``_resolve_start``): beside the slots' launches, on a context of its own, the chains of 100
    restarts stretched the slots' 0.74 s to 0.87 s (JRA-55-shaped problem)."""

    def __init__(self, n, p, dtype_code):
        self.n, self.p, self.dtype_code = n, p, dtype_code
        self.request = None

    def set_linear_kernel(self, on):
        pass

    def furthest_sum(self, n_components, start_index, exclude=None, extra_steps=1):
        self.request = (int(n_components), int(start_index), [int(e) for e in ([] if exclude is None else exclude)],
                        int(extra_steps))
        return np.arange(int(n_components), dtype=np.int64)


def _resolve_start(ctx, start, data, m):
    """The start factors of a restart with its FurthestSum selection made now, on ``ctx`` (same data,
    same dtype as the feed's draw context would have had: the same picks)."""
    request = start.get("selection")
    if request is None:
        return start
    from .archetypal_analysis import _furthest_sum_on_device, _one_hot_rows
    k, start_index, exclude, extra = request
    selected = _furthest_sum_on_device(ctx, data.shape[0], k, start_index, extra, np.asarray(exclude, dtype='i8'))
    start = dict(start)
    del start["selection"]
    if isinstance(m, ArchetypalAnalysis):
        start["dictionary"] = _one_hot_rows(selected, data.shape[0], np.float64)
    else:
        start["dictionary"] = np.ascontiguousarray(np.asarray(data, dtype=np.float64)[selected].T)
    return start


class _RestartFeed(object):
    """The models and starting factors of the restarts, produced in the drivers' order -- make_model(),
    then that model's draws from the shared ``RandomState``, restart after restart exactly as the
    sequential loop interleaves them -- by ONE worker thread, so that the device iterates on the first
    restarts while the host draws the later ones (a draw is 1-3 ms: the reference's RNG order asks for a
    fresh n x k weights matrix per restart, a FurthestSum start for a chain of device launches on a
    context of its own).  ``wait(i)`` blocks until restart i is there and re-raises what the producer
    raised."""

    def __init__(self, make_model, data, n_init):
        self.models = [None] * n_init
        self.starts = [None] * n_init
        self.same = [True] * n_init               # same class and hyper-parameters as the first restart
        self._ready = [threading.Event() for _ in range(n_init)]
        self._error = None
        self._make, self._data = make_model, data
        _backend.load_library()
        _backend.release_device_cache()
        self._thread = threading.Thread(target=self._produce, name="restart-draws")
        self._thread.daemon = True
        self._thread.start()

    def _produce(self):
        data, cache = self._data, {}              # data-dependent constants of the initialisers: |X| mean,
        draw_ctx = None                           # FurthestSum's distance columns; ONE context for all draws
        try:
            for i in range(len(self.models)):     # RNG draws in the sequential loop's order
                m = self._make()
                if not isinstance(m, (ArchetypalAnalysis, GPNHConvexCoding)):
                    raise TypeError("fit_restarts handles ArchetypalAnalysis and GPNHConvexCoding models")
                needs_device = (m.init == 'furthest_sum' or (m.init is None and isinstance(m, ArchetypalAnalysis)))
                # a selection that would run as a chain of device launches is left to whoever loads the
                # restart (_SelectionRecorder); one that reads distance columns out of the shared cache
                # (short and wide matrices: archetypal_analysis._furthest_sum_on_device) stays here
                code = _backend.dtype_code(m.dtype)
                column_bytes = float(data.shape[0]) * float(data.shape[1]) * (4 if code == _backend.AA_F32 else 8)
                recorder = None
                if (needs_device and _backend.distributed_env() is None and _aa_module._FURTHEST_SUM_ON_DEVICE
                        and column_bytes <= 64e6 and data.ndim == 2):
                    recorder = _SelectionRecorder(data.shape[0], data.shape[1], code)
                elif needs_device and draw_ctx is None and _backend.distributed_env() is None:
                    draw_ctx = _backend.Context(dtype=m.dtype)
                    draw_ctx.set_data(data)
                extra = {}
                if recorder is not None:
                    extra = dict(_draw_ctx=recorder)
                elif needs_device and draw_ctx is not None:
                    extra = dict(_draw_ctx=draw_ctx)
                if isinstance(m, ArchetypalAnalysis):
                    C0, Z0, a0 = m._aa(data, _draw_only=True, _cache=cache, **extra)
                    start = dict(dictionary=C0, weights=Z0, alpha=a0)
                else:
                    W0, Z0 = m._gpnh_convex_coding(data, _draw_only=True, _cache=cache, **extra)
                    start = dict(dictionary=W0, weights=Z0)
                if recorder is not None and recorder.request is not None:
                    start["selection"] = recorder.request      # its dictionary is a placeholder until then
                if i > 0:
                    self.same[i] = _same_settings(m, self.models[0])
                self.models[i], self.starts[i] = m, start
                self._ready[i].set()
        except BaseException as e:                # handed to whoever waits
            self._error = e
        finally:
            if draw_ctx is not None:
                draw_ctx.close()
            for ev in self._ready:
                ev.set()

    def wait(self, i):
        self._ready[i].wait()
        if self.models[i] is None:
            raise self._error
        return i

    def join(self):
        self._thread.join()
        if self._error is not None:
            raise self._error


class _FeedView(object):
    """models / starts of a share of the restarts as the slot loops index them: item j is restart idx[j],
    and asking for it waits until the feed has produced it."""

    def __init__(self, feed, idx, what):
        self._feed, self._idx, self._what = feed, idx, what

    def __len__(self):
        return len(self._idx)

    def __getitem__(self, j):
        i = self._feed.wait(self._idx[j])
        return (self._feed.models if self._what == "models" else self._feed.starts)[i]

    def same(self, j):
        i = self._feed.wait(self._idx[j])
        return self._feed.same[i]


def _next_pending(pending, models, left):
    """The next restart that can sit beside the first one; restarts with other settings go to `left`
    (the generic path runs them)."""
    while pending:
        i = pending.pop(0)
        if not hasattr(models, "same") or models.same(i):
            return i
        left.append(i)
    return None


def _fit_gpnh_slots(models, starts, data, device, poll_every=8, n_slots=None):
    """Up to 64 // k restarts side by side in ONE set of device arrays (aa_gpnh_slots_*): every outer
    iteration's launches serve all of them, a restart that stops is taken out and the next pending
    one takes its slot.  Restart by restart the result is the one the sequential loop gives, bit
    for bit (each slot runs the single fit's arithmetic on its own columns).  Returns the indices
    of restarts whose normal equations were not positive definite (the sequential path solves those
    with lstsq like the reference)."""
    import time
    import warnings
    m0 = models[0]
    k = m0.n_components
    n_samples = data.shape[0]
    n_slots = min(_gpnh_slot_count(m0, n_samples, n_slots), len(models))
    fallback, errors = [], {}
    ctx = _backend.Context(dtype=m0.dtype, device=device)
    try:
        ctx.set_data(data)
        mono_tol = m0.tolerance
        if ctx.dtype_code == _backend.AA_F32:
            mono_tol = max(m0.tolerance, 8 * 6e-8 * ctx.data_trace() / n_samples)
        ctx.gpnh_slots_begin(n_slots, k, m0.lambda_W, m0.max_iterations, m0.tolerance, m0.stopping_criterion,
                             m0.require_monotonic_cost_decrease, m0.weights_solver_kwargs, mono_tolerance=mono_tol)
        pending = list(range(len(models)))
        owner = [None] * n_slots
        loaded_at = [0.0] * n_slots
        tables = [None] * len(models)

        prof = slots_profile
        prof.update(load=0.0, run=0.0, fetch=0.0, polls=0, slots=n_slots)

        def load(r):
            i = _next_pending(pending, models, fallback)
            if i is None:
                return
            start = starts[i]                     # (waits for the draw if the feed is behind)
            t0 = time.perf_counter()
            start = _resolve_start(ctx, start, data, models[i])
            ctx.gpnh_slots_load(r, start["dictionary"], start["weights"])
            owner[r] = i
            loaded_at[r] = time.perf_counter()
            prof["load"] += loaded_at[r] - t0

        for r in range(n_slots):
            load(r)
        while any(o is not None for o in owner):
            t0 = time.perf_counter()
            status = ctx.gpnh_slots_run(poll_every)
            prof["run"] += time.perf_counter() - t0
            prof["polls"] += 1
            for r, st in enumerate(status):
                i = owner[r]
                if i is None:
                    continue
                if st.not_spd:
                    fallback.append(i)
                elif st.stop:
                    m = models[i]
                    if st.error_stage:
                        errors[i] = RuntimeError('factorization cost increased after {} update'.format(
                            'dictionary' if st.error_stage == 1 else 'weights'))
                    else:
                        t0 = time.perf_counter()
                        Z, W, cost0, costs = ctx.gpnh_slots_fetch(r, st.stop_iter)
                        prof["fetch"] += time.perf_counter() - t0
                        finals = costs[1::2]
                        begins = np.concatenate(([cost0], finals[:-1]))
                        m.weights, m.dictionary = Z, W
                        m.cost, m.n_iter = float(finals[-1]), int(st.stop_iter)
                        m.cost_deltas = [float(d) for d in finals - begins]
                        m.avg_time_per_iter = (time.perf_counter() - loaded_at[r]) / max(st.iterations_run, 1)
                        tables[i] = (finals, begins, m.avg_time_per_iter, bool(st.converged))
                        if m.n_iter == m.max_iterations and m.tolerance > 0:
                            warnings.warn('Maximum number of iterations %d reached.' % m.max_iterations, UserWarning)
                else:
                    continue
                owner[r] = None
                if pending:
                    load(r)
        ctx.aa_slots_end()
    finally:
        ctx.close()
    if m0.verbose:
        _print_tables(models, tables, "*** GPNH convex coding: n_components = {:d} ***", 100)
    if errors:
        raise errors[min(errors)]
    return fallback


def _aa_slots_eligible(m0, n_models, data):
    """AA restarts that can share one set of device arrays (aa_slots_*): the drivers' setting -- one SPG
    iteration per dictionary update -- fewer than 65 536 samples, k <= 16 (and every restart with the
    first one's hyper-parameters: _same_settings)."""
    if type(m0) is not ArchetypalAnalysis or n_models < 2:
        return False
    k = m0.n_components
    dkw = dict(m0.dictionary_solver_kwargs)
    return (isinstance(k, int) and 1 <= k <= 16 and data.shape[0] < 65536
            and m0.scale_factors_solver_kwargs.get("memory", 10) <= 16
            and dkw.get("max_iterations", 1000) == 1 and dkw.get("memory", 1) <= 16
            and m0.weights_solver_kwargs.get("memory", 1) <= 1
            and m0.weights_solver_kwargs.get("max_iterations", 1000) >= 1
            and _backend.distributed_env() is None)


def _fit_aa_slots(models, starts, data, device, poll_every=8, n_slots=None):
    """Up to 32 // k AA restarts side by side in ONE set of device arrays (aa_slots_*): every launch of
    an outer iteration serves all of them; a restart that stops is taken out and the next pending one
    takes its slot (its first dictionary update is the cold one of a fit while the others carry on:
    aa_slots_reload).  Restart by restart the result is the sequential loop's, bit for bit."""
    import time
    import warnings
    from .archetypal_analysis import _warn_from_spg_flags, _DEVICE_LOOP_BATCH

    class _Flags(object):
        def __init__(self, flags):
            self.spg_flags = flags

    m0 = models[0]
    k = m0.n_components
    n_slots = min((32 // k) if n_slots is None else int(n_slots), 32 // k, len(models))
    n_samples = data.shape[0]
    prof = slots_profile
    prof.update(load=0.0, run=0.0, fetch=0.0, polls=0, slots=n_slots)
    ctx = _backend.Context(dtype=m0.dtype, device=device)
    errors, left = {}, []
    try:
        ctx.set_data(data)
        mono_tol = m0.tolerance
        if ctx.dtype_code == _backend.AA_F32:
            mono_tol = max(m0.tolerance, 8 * 6e-8 * ctx.data_trace() / n_samples)
        pending = list(range(len(models)))
        owner = [None] * n_slots
        loaded_at = [0.0] * n_slots
        tables = [None] * len(models)
        t0 = time.perf_counter()
        ctx.aa_slots_begin(n_slots, k, m0.max_iterations, m0.tolerance, m0.stopping_criterion,
                           m0.require_monotonic_cost_decrease, m0.dictionary_solver_kwargs,
                           m0.weights_solver_kwargs, mono_tolerance=mono_tol, delta=m0.delta,
                           scale_kw=m0.scale_factors_solver_kwargs)
        for r in range(n_slots):                  # the first group starts together
            i = _next_pending(pending, models, left)
            if i is None:
                break
            start = _resolve_start(ctx, starts[i], data, models[i])
            ctx.aa_slots_load(r, start["dictionary"], start["weights"], start["alpha"])
            owner[r] = i
            loaded_at[r] = time.perf_counter()
        prof["load"] += time.perf_counter() - t0
        while any(o is not None for o in owner):
            t0 = time.perf_counter()
            status = ctx.aa_slots_run(poll_every)
            prof["run"] += time.perf_counter() - t0
            prof["polls"] += 1
            for r, st in enumerate(status):
                i = owner[r]
                if i is None or not st.stop:
                    continue
                m = models[i]
                t0 = time.perf_counter()
                _warn_from_spg_flags(_Flags(st.not_spd))
                if st.error_stage:
                    errors[i] = RuntimeError('factorization cost increased after {} update'.format(
                        {1: 'dictionary', 2: 'weights', 3: 'scale factors'}[st.error_stage]))
                else:
                    # aa_iterate rebuilds the products only when it ran past the stopping iteration
                    carried = (st.stop_iter + 1) % _DEVICE_LOOP_BATCH == 0 or st.stop_iter + 1 == m.max_iterations
                    Z, C, CX, cost0, costs, alpha = ctx.aa_slots_fetch(r, st.stop_iter, carried)
                    finals = costs[1::2]
                    begins = np.concatenate(([cost0], finals[:-1]))
                    if m.delta != 0:                  # ArchetypalAnalysis.fit_transform (reference :1140-1144)
                        C = np.dot(np.diag(alpha), C)
                        CX = alpha[:, np.newaxis] * CX
                    m.weights, m.dictionary, m.alpha = Z, C, alpha
                    m.cost, m.n_iter = float(finals[-1]), int(st.stop_iter)
                    m.cost_deltas = [d for d in finals - begins]
                    m.avg_time_per_iter = (time.perf_counter() - loaded_at[r]) / max(st.iterations_run, 1)
                    tables[i] = (finals, begins, m.avg_time_per_iter, bool(st.converged))
                    m.archetypes = CX
                    if m.n_iter == m.max_iterations and m.tolerance > 0:
                        warnings.warn('Maximum number of iterations %d reached.' % m.max_iterations, UserWarning)
                prof["fetch"] += time.perf_counter() - t0
                owner[r] = None
                i = _next_pending(pending, models, left)
                if i is not None:
                    start = starts[i]             # (waits for the draw if the feed is behind)
                    t0 = time.perf_counter()
                    start = _resolve_start(ctx, start, data, models[i])
                    ctx.aa_slots_reload(r, start["dictionary"], start["weights"], start["alpha"])
                    owner[r] = i
                    loaded_at[r] = time.perf_counter()
                    prof["load"] += loaded_at[r] - t0
        ctx.aa_slots_end()
    finally:
        ctx.close()
    if m0.verbose:
        _print_tables(models, tables, "*** AA: n_components = {:d} ***", 80)
    if errors:
        raise errors[min(errors)]
    return left


def fit_restarts(make_model, data, n_init, n_jobs=None, devices=None, side_by_side=True, n_slots=None):
    """``make_model()`` returns a fresh ``ArchetypalAnalysis`` or ``GPNHConvexCoding`` (the drivers
    pass the shared ``RandomState`` as its ``random_state``, ``init`` 'random' or 'furthest_sum').
    ``n_jobs``: worker threads (default: one per device -- more than one fit at a time on a GPU has
    not been faster).
    ``devices``: GPU indices the worker threads are dealt over (default: the current one); the data
    matrix is uploaded ONCE per device and the workers of a device share that copy
    (``aa_share_data``), each with its own factors, streams and scratch.
    ``side_by_side`` (models with the same hyper-parameters; GPNH with k <= 16, AA with the drivers'
    settings -- one SPG iteration per dictionary update, fewer than 65 536 samples, k <= 16):
    the restarts run ``n_slots`` at a time (default: 64 // k for GPNH, 32 // k for AA) in ONE set of
    device arrays per device and share every launch of an outer iteration (``_fit_gpnh_slots``,
    ``_fit_aa_slots``); with several ``devices`` restart i runs on device i mod G; ``n_jobs`` is not
    used then.
    Returns ``(models, best)``: the fitted models in restart order and the index of the first one
    with the lowest cost (the model the drivers' ``if cost < best_cost`` loop keeps)."""
    data = np.asarray(data)
    feed = _RestartFeed(make_model, data, n_init)     # models and draws in the sequential loop's order, on a thread
    devices = [_backend.device_index()] if devices is None else [int(d) for d in devices]
    todo = list(range(n_init))
    try:
        feed.wait(0)
        m0 = feed.models[0]
        slot_fit = None
        if side_by_side and _aa_slots_eligible(m0, n_init, data):
            slot_fit = _fit_aa_slots          # AA: groups of restarts share every launch (aa_slots_*)
        elif side_by_side and _slots_eligible(m0, n_init, data):
            slot_fit = _fit_gpnh_slots        # GPNH: slots refilled as restarts stop (aa_gpnh_slots_*)
        if slot_fit is not None:
            # the restarts are dealt over the devices (restart i on device i mod G), every device runs its
            # share side by side on its own copy of the data; no collective anywhere.  The device iterates on
            # the first restarts while the feed draws the later ones.
            shares = [list(range(d, n_init, len(devices))) for d in range(len(devices))]
            shares = [sh for sh in shares if sh]

            def on_device(d):
                idx = shares[d]
                left = slot_fit(_FeedView(feed, idx, "models"), _FeedView(feed, idx, "starts"), data, devices[d],
                                n_slots=n_slots)
                return [idx[j] for j in left]

            if len(shares) == 1:
                todo = on_device(0)
            else:
                with ThreadPoolExecutor(max_workers=len(shares)) as pool:
                    todo = sorted(i for left in pool.map(on_device, range(len(shares))) for i in left)
    except BaseException:
        feed._thread.join()                       # the producer closes its context before the error travels on
        raise
    feed.join()                                   # every draw made (or the producer's error raised)
    models, starts = feed.models, feed.starts
    if not todo:
        costs = [m.cost for m in models]
        return models, int(np.argmin(costs))
    _backend.release_device_cache()               # the workers bring their own contexts
    local = threading.local()
    lock = threading.Lock()
    owners = {}                                   # device -> context that holds the data matrix
    sharers = []
    n_workers = [0]

    def run(i):
        if not hasattr(local, "dd"):
            with lock:
                dev = devices[n_workers[0] % len(devices)]
                n_workers[0] += 1
                if dev not in owners:             # first worker of a device: uploads, and works on it
                    ctx = _backend.Context(dtype=models[i].dtype, device=dev)
                    ctx.set_data(data)
                    owners[dev] = ctx
                else:
                    ctx = _backend.Context(dtype=models[i].dtype, device=dev)
                    ctx.share_data(owners[dev])
                    sharers.append(ctx)
            local.dd = DeviceData(ctx, data.shape, None, data.shape[1:])
        m = models[i]
        init = m.init
        m.init = 'custom'
        try:
            m.fit_transform(local.dd, **_resolve_start(local.dd._ctx, starts[i], data, m))
        finally:
            m.init = init
        return m.cost

    if n_jobs is None:
        # one worker per device: several fits at a time on ONE GPU are not faster (their launch chains
        # do not overlap: profiles/round3_restarts_threads.txt, round3_stream_interleave.txt)
        n_jobs = len(devices)
    try:
        with ThreadPoolExecutor(max_workers=max(1, int(n_jobs))) as pool:
            list(pool.map(run, todo))
        costs = [m.cost for m in models]
    finally:
        for ctx in sharers:                       # the aliases first, then the owners
            ctx.close()
        for ctx in owners.values():
            ctx.close()
    best = int(np.argmin(costs))                  # first minimum, like `if cost < best_cost`
    return models, best
