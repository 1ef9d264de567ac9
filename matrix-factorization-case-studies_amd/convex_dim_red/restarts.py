"""The drivers' restart loops, several fits at a time.

``bin/run_hadisst_aa.py:149-174`` / ``bin/run_jra55_pca_gpnh.py:112-138`` fit ``n_init`` fresh
models one after the other, all drawing their starting factors from ONE shared ``RandomState``,
and keep the one with the lowest cost.  For the problems those drivers run (161 MB / 15 MB
matrices) one fit leaves most of an MI355X idle -- an outer iteration is a few dozen dependent
launches of a few microseconds each -- so ``fit_restarts`` draws the starting factors of all
restarts first, in the drivers' order (so every restart starts exactly where it would in the
sequential loop), and then

* GPNH models, and AA models with the drivers' settings (one SPG iteration per dictionary update,
  any delta, fewer than 65 536 samples, k <= 16): lays the restarts SIDE BY SIDE in the component slots of
  one set of device arrays, where they share every launch of an outer iteration; a restart that stops
  hands its slot to the next one (``_fit_gpnh_slots`` / aa_gpnh_slots_*, ``_fit_aa_slots`` /
  aa_slots_*): 3.5-5x the sequential loop's speed on the JRA-55- and HadISST-shaped problems with
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


def _slots_eligible(models, data):
    """GPNH restarts that can share one set of device arrays: same hyper-parameters, at least four
    restarts of k components in the 64 component slots of the tall arrays.  Mirrors the
    preconditions of aa_gpnh_slots_begin (csrc/solver.hip): QPs of more than four passes take the
    AA slots' four-lane / wave-per-sample launch, which exists for fewer than 65 536 samples."""
    m0 = models[0]
    if not all(isinstance(m, GPNHConvexCoding) for m in models) or len(models) < 2:
        return False
    keys = ("n_components", "lambda_W", "tolerance", "max_iterations", "stopping_criterion", "verbose",
            "require_monotonic_cost_decrease", "dtype", "weights_solver_kwargs", "dictionary_solver_kwargs")
    if any(getattr(m, a) != getattr(m0, a) for m in models[1:] for a in keys):
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
            i = pending.pop(0)
            t0 = time.perf_counter()
            ctx.gpnh_slots_load(r, starts[i]["dictionary"], starts[i]["weights"])
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


def _aa_slots_eligible(models, data):
    """AA restarts that can share one set of device arrays (aa_slots_*): the drivers' setting -- one SPG
    iteration per dictionary update -- same hyper-parameters, fewer than 65 536 samples, k <= 16."""
    m0 = models[0]
    if not all(type(m) is ArchetypalAnalysis for m in models) or len(models) < 2:
        return False
    keys = ("n_components", "delta", "tolerance", "max_iterations", "stopping_criterion", "verbose",
            "require_monotonic_cost_decrease", "dtype", "weights_solver_kwargs", "dictionary_solver_kwargs",
            "scale_factors_solver_kwargs")
    if any(getattr(m, a) != getattr(m0, a) for m in models[1:] for a in keys):
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
    errors = {}
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
            i = pending.pop(0)
            ctx.aa_slots_load(r, starts[i]["dictionary"], starts[i]["weights"], starts[i]["alpha"])
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
                if pending:
                    t0 = time.perf_counter()
                    i = pending.pop(0)
                    ctx.aa_slots_reload(r, starts[i]["dictionary"], starts[i]["weights"], starts[i]["alpha"])
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
    return []


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
    models, starts = [], []
    cache = {}                                    # data-dependent constants of the initialisers: |X| mean,
                                                  # FurthestSum's distance columns (the same matrix every time)
    draw_ctx = None                               # FurthestSum starts: ONE context for all draws
    try:
        for _ in range(n_init):                   # RNG draws in the sequential loop's order
            m = make_model()
            if not isinstance(m, (ArchetypalAnalysis, GPNHConvexCoding)):
                raise TypeError("fit_restarts handles ArchetypalAnalysis and GPNHConvexCoding models")
            needs_device = (m.init == 'furthest_sum' or (m.init is None and isinstance(m, ArchetypalAnalysis)))
            if needs_device and draw_ctx is None and _backend.distributed_env() is None:
                _backend.release_device_cache()
                draw_ctx = _backend.Context(dtype=m.dtype)
                draw_ctx.set_data(data)
            extra = dict(_draw_ctx=draw_ctx) if (needs_device and draw_ctx is not None) else {}
            if isinstance(m, ArchetypalAnalysis):
                C0, Z0, a0 = m._aa(data, _draw_only=True, _cache=cache, **extra)
                starts.append(dict(dictionary=C0, weights=Z0, alpha=a0))
            else:
                W0, Z0 = m._gpnh_convex_coding(data, _draw_only=True, _cache=cache, **extra)
                starts.append(dict(dictionary=W0, weights=Z0))
            models.append(m)
    finally:
        if draw_ctx is not None:
            draw_ctx.close()
    _backend.release_device_cache()               # the workers bring their own contexts
    devices = [_backend.device_index()] if devices is None else [int(d) for d in devices]
    todo = list(range(n_init))
    slot_fit = None
    if side_by_side and _aa_slots_eligible(models, data):
        slot_fit = _fit_aa_slots          # AA: groups of restarts share every launch (aa_slots_*)
    elif side_by_side and _slots_eligible(models, data):
        slot_fit = _fit_gpnh_slots        # GPNH: slots refilled as restarts stop (aa_gpnh_slots_*)
    if slot_fit is not None:
        # the restarts are dealt over the devices (restart i on device i mod G), every device runs its
        # share side by side on its own copy of the data; no collective anywhere
        shares = [list(range(d, n_init, len(devices))) for d in range(len(devices))]
        shares = [sh for sh in shares if sh]

        def on_device(d):
            idx = shares[d]
            left = slot_fit([models[i] for i in idx], [starts[i] for i in idx], data, devices[d], n_slots=n_slots)
            return [idx[j] for j in left]

        if len(shares) == 1:
            todo = on_device(0)
        else:
            with ThreadPoolExecutor(max_workers=len(shares)) as pool:
                todo = sorted(i for left in pool.map(on_device, range(len(shares))) for i in left)
        if not todo:
            costs = [m.cost for m in models]
            return models, int(np.argmin(costs))
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
            m.fit_transform(local.dd, **starts[i])
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
