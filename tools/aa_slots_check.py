#!/usr/bin/env python
"""AA restarts side by side (fit_restarts: aa_slots_*) against the sequential loop of the drivers on
the C2 stand-in (HadISST shape: 1610 x 25 000, k = 5, random starts, one SPG iteration per dictionary
update): per restart the same cost, n_iter, cost deltas, weights, dictionary and archetypes; wall clock."""
import os
import sys
import time
import warnings

import numpy as np

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd"))
sys.path.insert(0, _R)
import convex_dim_red as cdr  # noqa: E402
from convex_dim_red import _backend, restarts as _rs  # noqa: E402
from oracle import aa_oracle as orc  # noqa: E402

warnings.simplefilter("ignore")
n_init = int(sys.argv[1]) if len(sys.argv) > 1 else 12
small = len(sys.argv) > 2 and sys.argv[2] == "small"
n, p, k = (600, 300, 5) if small else (1610, 25000, 5)
rng = np.random.RandomState(0)
B = rng.standard_normal((k, p))
Zt = orc.right_stochastic_matrix((n, k), rng) ** 4
Zt /= Zt.sum(axis=1, keepdims=True)
X = Zt.dot(B) + 0.05 * rng.standard_normal((n, p))

for dtype, init in (("float64", "random"), ("float32", "random"), ("float64", "furthest_sum")):
    Xd = X.astype(np.float32) if dtype == "float32" else X

    def make(rs):
        return cdr.ArchetypalAnalysis(k, init=init, tolerance=1e-4, max_iterations=10000, random_state=rs, dtype=dtype,
                                      dictionary_solver_kwargs=dict(max_iterations=1))

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
    shared = np.random.RandomState(0)
    t0 = time.perf_counter()
    models, best = cdr.fit_restarts(lambda: make(shared), Xd, n_init)
    t_sbs = time.perf_counter() - t0
    same = [a.cost == b.cost and a.n_iter == b.n_iter and np.array_equal(a.weights, b.weights)
            and np.array_equal(a.dictionary, b.dictionary) and list(a.cost_deltas) == list(b.cost_deltas)
            and np.array_equal(a.archetypes, b.archetypes) for a, b in zip(seq, models)]
    its = sum(m.n_iter + 1 for m in seq)
    print("%s init=%s: n_init=%d, %d outer iterations (per restart %d..%d): sequential %.3f s, side by side %.3f s = %.2fx; "
          "identical per restart: %s; best %d / %d"
          % (dtype, init, n_init, its, min(m.n_iter for m in seq) + 1, max(m.n_iter for m in seq) + 1, t_seq, t_sbs,
             t_seq / t_sbs, all(same), best, int(np.argmin([m.cost for m in seq]))), flush=True)
    print("      side by side: %d slots, %d polls of 8 iterations, load %.3f s, iterate %.3f s, fetch %.3f s"
          % (_rs.slots_profile.get("slots", 0), _rs.slots_profile.get("polls", 0), _rs.slots_profile.get("load", 0),
             _rs.slots_profile.get("run", 0), _rs.slots_profile.get("fetch", 0)), flush=True)
    if not all(same):
        for i, (a, b) in enumerate(zip(seq, models)):
            nd = min(len(a.cost_deltas), len(b.cost_deltas))
            dd = np.abs(np.asarray(a.cost_deltas[:nd]) - np.asarray(b.cost_deltas[:nd]))
            first = int(np.argmax(dd > 0)) if (dd > 0).any() else -1
            print("   restart %d: cost %.15g / %.15g, n_iter %d / %d, max |dZ| %.2e, max |dC| %.2e, max |dCX| %.2e, first differing cost delta at %d"
                  % (i, a.cost, b.cost, a.n_iter, b.n_iter, np.abs(a.weights - b.weights).max(),
                     np.abs(a.dictionary - b.dictionary).max(), np.abs(a.archetypes - b.archetypes).max(), first))
        sys.exit(1)
print("AA_SLOTS_OK")
