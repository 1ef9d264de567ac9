"""The drivers' restart loops, several fits at a time.

``bin/run_hadisst_aa.py:149-174`` / ``bin/run_jra55_pca_gpnh.py:112-138`` fit ``n_init`` fresh
models one after the other, all drawing their starting factors from ONE shared ``RandomState``,
and keep the one with the lowest cost.  For the problems those drivers run (161 MB / 15 MB
matrices) one fit leaves most of an MI355X idle -- an outer iteration is a few dozen dependent
launches of a few microseconds each -- so ``fit_restarts`` draws the starting factors of all
restarts first, in the drivers' order (so every restart starts exactly where it would in the
sequential loop), and then runs the fits ``n_jobs`` at a time, each worker thread on its own
device context; the contexts of a device share one resident copy of the data.  Every model ends
with the attributes the sequential loop gives it; results are identical, restart by restart.
"""
from __future__ import absolute_import, division

import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _backend
from .archetypal_analysis import ArchetypalAnalysis
from .gpnh_convex_coding import GPNHConvexCoding
from .preprocessing import DeviceData


def fit_restarts(make_model, data, n_init, n_jobs=4, devices=None):
    """``make_model()`` returns a fresh ``ArchetypalAnalysis`` or ``GPNHConvexCoding`` (the drivers
    pass the shared ``RandomState`` as its ``random_state``, ``init`` 'random' or 'furthest_sum').
    ``devices``: GPU indices the worker threads are dealt over (default: the current one); the data
    matrix is uploaded ONCE per device and the workers of a device share that copy
    (``aa_share_data``), each with its own factors, streams and scratch.
    Returns ``(models, best)``: the fitted models in restart order and the index of the first one
    with the lowest cost (the model the drivers' ``if cost < best_cost`` loop keeps)."""
    data = np.asarray(data)
    models, starts = [], []
    for _ in range(n_init):                       # RNG draws in the sequential loop's order
        m = make_model()
        if isinstance(m, ArchetypalAnalysis):
            C0, Z0, a0 = m._aa(data, _draw_only=True)
            starts.append(dict(dictionary=C0, weights=Z0, alpha=a0))
        elif isinstance(m, GPNHConvexCoding):
            W0, Z0 = m._gpnh_convex_coding(data, _draw_only=True)
            starts.append(dict(dictionary=W0, weights=Z0))
        else:
            raise TypeError("fit_restarts handles ArchetypalAnalysis and GPNHConvexCoding models")
        models.append(m)
    _backend.release_device_cache()               # the workers bring their own contexts
    devices = [_backend.device_index()] if devices is None else [int(d) for d in devices]
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

    try:
        with ThreadPoolExecutor(max_workers=max(1, int(n_jobs))) as pool:
            costs = list(pool.map(run, range(n_init)))
    finally:
        for ctx in sharers:                       # the aliases first, then the owners
            ctx.close()
        for ctx in owners.values():
            ctx.close()
    best = int(np.argmin(costs))                  # first minimum, like `if cost < best_cost`
    return models, best
