#!/usr/bin/env python
"""GPNH restarts side by side (fit_restarts(side_by_side=True): aa_gpnh_slots_*) against the
sequential loop of the drivers on the C3 stand-in: per restart the same cost, n_iter, cost deltas,
weights and dictionary; wall clock of n_init restarts both ways."""
import os
import sys
import time
import warnings

import numpy as np

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd"))
sys.path.insert(0, _R)
import convex_dim_red as cdr  # noqa: E402
from convex_dim_red import _backend  # noqa: E402
from oracle import aa_oracle as orc  # noqa: E402

warnings.simplefilter("ignore")
n_init = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n, p, k = 22280, 167, 10
rng = np.random.RandomState(0)
W0 = rng.standard_normal((p, k))
Zt = orc.right_stochastic_matrix((n, k), rng)
X = Zt.dot(W0.T) + 0.1 * rng.standard_normal((n, p))

for dtype, lam, init in (("float64", 0, "random"), ("float64", 1, "random"), ("float32", 0, "random"),
                         ("float64", 0, "furthest_sum")):
    Xd = X.astype(np.float32) if dtype == "float32" else X

    def make(rs):
        return cdr.GPNHConvexCoding(k, lambda_W=lam, init=init, tolerance=1e-6, max_iterations=10000, random_state=rs,
                                    stopping_criterion="rel_delta_f", dtype=dtype,
                                    weights_solver_kwargs=dict(max_iterations=1))

    _backend.release_device_cache()
    make(np.random.RandomState(5)).fit_transform(Xd[:64])     # warm the library
    _backend.release_device_cache()
    shared = np.random.RandomState(0)
    t0 = time.perf_counter()
    seq = []
    for _ in range(n_init):
        m = make(shared)
        m.fit_transform(Xd)
        seq.append(m)
    t_seq = time.perf_counter() - t0
    outs = {}
    for sbs in (3, True, False):
        shared = np.random.RandomState(0)
        t0 = time.perf_counter()
        models, best = cdr.fit_restarts(lambda: make(shared), Xd, n_init, n_jobs=2, side_by_side=bool(sbs),
                                        n_slots=3 if sbs == 3 else None)
        outs[sbs] = (time.perf_counter() - t0, models, best)
        if sbs == 3:
            print("      three slots: %.3f s" % outs[sbs][0], flush=True)
    t_sbs, models, best = outs[True]
    same = all(a.cost == b.cost and a.n_iter == b.n_iter and np.array_equal(a.weights, b.weights)
               and np.array_equal(a.dictionary, b.dictionary) and list(a.cost_deltas) == list(b.cost_deltas)
               for a, b in zip(seq, models))
    its = sum(m.n_iter + 1 for m in seq)
    print("%s lambda=%g init=%s: n_init=%d, %d outer iterations (per restart %d..%d): sequential %.3f s, threads (n_jobs=2) %.3f s, "
          "side by side %.3f s = %.2fx; identical per restart: %s; best %d / %d"
          % (dtype, lam, init, n_init, its, min(m.n_iter for m in seq) + 1, max(m.n_iter for m in seq) + 1, t_seq,
             outs[False][0], t_sbs, t_seq / t_sbs, same, best, int(np.argmin([m.cost for m in seq]))), flush=True)
    from convex_dim_red import restarts as _rs
    print("      side by side: %d slots, %d polls of 8 iterations, load %.3f s, iterate %.3f s, fetch %.3f s"
          % (_rs.slots_profile["slots"], _rs.slots_profile["polls"], _rs.slots_profile["load"], _rs.slots_profile["run"],
             _rs.slots_profile["fetch"]), flush=True)
    if not same:
        for i, (a, b) in enumerate(zip(seq, models)):
            print("   restart %d: cost %.15g / %.15g, n_iter %d / %d, max |dZ| %.2e, max |dW| %.2e"
                  % (i, a.cost, b.cost, a.n_iter, b.n_iter, np.abs(a.weights - b.weights).max(),
                     np.abs(a.dictionary - b.dictionary).max()))
        sys.exit(1)
print("SLOTS_OK")
