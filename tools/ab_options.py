#!/usr/bin/env python
"""A/B timing of library options on the headline problem: for every option set given on the
command line ("name=value,name=value" per argument; "" = defaults) the state is reset to
bench.py's start, 5 outer iterations warm up and 20 (the driver's window) + 30 more are timed.
Repeated twice, interleaved."""
import os
import sys
import time

import numpy as np

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_R, "matrix-factorization-case-studies_amd"))
sys.path.insert(0, _R)
import bench  # noqa: E402
from convex_dim_red import _backend  # noqa: E402

n, p, k = int(os.environ.get("N", bench.N_SAMPLES)), bench.N_FEATURES, bench.N_COMPONENTS
sets = sys.argv[1:] or [""]
X = bench.synthetic_rows(0, n)
C0, Z0 = bench.start_factors(n, k)
dkw = dict(max_iterations=1)
with _backend.Context(dtype=os.environ.get("DTYPE", "float32")) as ctx:
    ctx.set_data(X)
    for rep in range(2):
        for spec in sets:
            opts = dict(item.split("=") for item in spec.split(",") if item)
            for name, v in opts.items():
                _backend.set_option(name, int(v))
            ctx.set_state(C0, Z0, np.ones(k))
            ctx.prepare()
            ctx.outer_iterations(5, dkw, {})
            t0 = time.perf_counter()
            c1 = ctx.outer_iterations(20, dkw, {})
            t1 = time.perf_counter()
            c2 = ctx.outer_iterations(30, dkw, {})
            t2 = time.perf_counter()
            print("%-40s iterations 5-25: %.3f ms (%.1f it/s)   25-55: %.3f ms   cost %.9f -> %.9f" %
                  (spec or "(defaults)", 1e3 * (t1 - t0) / 20, 20 / (t1 - t0), 1e3 * (t2 - t1) / 30, c1[-1], c2[-1]), flush=True)
            for name in opts:
                _backend.set_option(name, {"qp_overlap_tail": 0, "row_local_acc64": 1, "qp_quad_cap": 0,
                                            "fuse_finalize": 1, "qp_wave_blocks": 1024, "qp_tail_cap": 96, "proj_res_side": 1, "qp_live_blocks": 48, "qp_quad_occ": 3, "qp_live_occ": 3}.get(name, 0))
