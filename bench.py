#!/usr/bin/env python
"""Headline benchmark: archetypal-analysis outer iterations per second on the synthetic
fp32 100000 x 4096, k = 32 problem (BASELINE.json configs[3]/[4]).

    python bench.py --gpus N --steps K --warmup W

A "step" is ONE outer iteration of the alternating solver on data already resident in
HBM: dictionary SPG update (reference archetypal_analysis.py:324-341 with the
production setting dictionary_solver_kwargs={max_iterations: 1}) + Gram refresh
(:618-627) + per-sample simplex-QP weights update (:369-396, default solver settings)
+ Gram refresh (:640-648).  N > 1 (launched by torch.distributed.run, one rank per
GPU): X is row-sharded, the total problem is fixed (strong scaling), RCCL all-reduces
only the k x p / k x k Gram products and packed scalars.

No PyTorch here: ranks read RANK/LOCAL_RANK/WORLD_SIZE/MASTER_PORT from the
environment, exchange the RCCL unique id through a file in /tmp, and do the barrier and
the max-over-ranks through the library's own communicator.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "matrix-factorization-case-studies_amd"))
sys.path.insert(0, ROOT)

N_SAMPLES, N_FEATURES, N_COMPONENTS = 100000, 4096, 32
BLOCK_ROWS = 500                      # synthetic data is generated in seeded row blocks
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: dense fp32 matrix peak


def synthetic_rows(lo, hi, n=N_SAMPLES, p=N_FEATURES, k=N_COMPONENTS):
    """Rows [lo, hi) of the benchmark matrix X = Zt B + 0.05 noise (float32).  Archetypes
    B from RandomState(0); every BLOCK_ROWS-row block of (Zt, noise) from its own
    RandomState(1000 + block), so the matrix does not depend on how it is sharded."""
    B = np.random.RandomState(0).standard_normal((k, p))
    out = np.empty((hi - lo, p), dtype=np.float32)
    b0, b1 = lo // BLOCK_ROWS, (hi + BLOCK_ROWS - 1) // BLOCK_ROWS
    for blk in range(b0, b1):
        r0, r1 = blk * BLOCK_ROWS, min((blk + 1) * BLOCK_ROWS, n)
        rng = np.random.RandomState(1000 + blk)
        Zt = rng.uniform(size=(r1 - r0, k))
        Zt /= Zt.sum(axis=1, keepdims=True)
        Zt = Zt ** 4                                      # peaky weights: identifiable hull
        Zt /= Zt.sum(axis=1, keepdims=True)
        rows = Zt.dot(B) + 0.05 * rng.standard_normal((r1 - r0, p))
        s0, s1 = max(lo, r0), min(hi, r1)
        out[s0 - lo:s1 - lo] = rows[s0 - r0:s1 - r0]
    return out


def start_factors(n=N_SAMPLES, k=N_COMPONENTS):
    """init='random' start (reference archetypal_analysis.py:51-70) from RandomState(1)."""
    rs = np.random.RandomState(1)
    C = rs.uniform(size=(k, n))
    C /= C.sum(axis=1, keepdims=True)
    Z = rs.uniform(size=(n, k))
    Z /= Z.sum(axis=1, keepdims=True)
    return C, Z


def exchange_unique_id(rank, world, backend):
    """RCCL unique id of this launch's communicator (rank 0 publishes it through a file)."""
    return backend.exchange_unique_id(rank, world, "bench")


def cpu_baseline(k, p, spg_kw, n_sample, steps):
    """The reference's op sequence (oracle.iterate_aa: 11 GEMM passes over X, sort-based
    projections, serial per-sample QPs) in float64 on the host, on the first n_sample
    rows of the workload from the same kind of start; returns (it/s scaled to the full n,
    description, final cost on the sample, GPU-comparable inputs)."""
    from oracle import aa_oracle as orc
    X = synthetic_rows(0, n_sample).astype(np.float64)
    C, Z = start_factors(n_sample, k)
    trace = float((X * X).sum())
    kw = dict(tolerance=0, dictionary_solver_kwargs=spg_kw, require_monotonic_cost_decrease=False,
              trace_XXt=trace)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Z1, C1, _, _, _, _, _ = orc.iterate_aa(X, Z, C, np.ones(k), max_iterations=1, **kw)   # warm-up
        t0 = time.perf_counter()
        timings = {}
        Z2, C2, _, cost, _, _, deltas = orc.iterate_aa(X, Z1, C1, np.ones(k), max_iterations=steps,
                                                       timings=timings, **kw)
        dt = time.perf_counter() - t0
    its = steps / dt
    return dict(its_sample=its, seconds=dt, cost=cost, timings=timings, X=X, C=C, Z=Z,
                used_c=orc.clib() is not None)


def converged_parity(k, p, dtype, device, spg_kw, qp_kw, n_small=1500, n_outer=250):
    """Near-convergence comparison (the north star's 'reconstruction error within 1e-5 rel
    of the NumPy reference'): oracle (float64 CPU, reference op sequence) and the HIP path
    run the SAME fixed number of outer iterations (no stopping test: a |delta cost| rule
    stops the two runs at different iterations on a flat cost curve) from the same start on
    the first n_small rows of the workload; the reconstruction errors are then compared in
    residual form, computed on the host in float64 for both.  The HIP path runs in the bench's
    arithmetic and in float64 (the reference dtype).  tests/test_gpu_longrun.py pins the same
    comparison under pytest against the committed oracle run (tests/golden/converged_1500.npz),
    whose own sensitivity to a one-ulp perturbation of X (`oracle_twin_rel`) is the yardstick."""
    import warnings
    from convex_dim_red import archetypal_analysis as aa
    from oracle import aa_oracle as orc
    X = synthetic_rows(0, n_small).astype(np.float64)
    C, Z = start_factors(n_small, k)
    trace = float((X * X).sum())
    kw = dict(tolerance=0, max_iterations=n_outer, dictionary_solver_kwargs=spg_kw,
              weights_solver_kwargs=qp_kw, require_monotonic_cost_decrease=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t0 = time.perf_counter()
        oZ, oC, _, ocost, oit, _, _ = orc.iterate_aa(X, Z, C, np.ones(k), trace_XXt=trace, **kw)
        t_cpu = time.perf_counter() - t0
    rec_o = 0.5 * np.linalg.norm(X - oZ.dot(oC.dot(X))) ** 2 / n_small

    def leg(dt):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            Xh = X.astype(np.float32) if dt == "float32" else X
            t0 = time.perf_counter()
            hZ, hC, _, hcost, hit, _, _ = aa._iterate_aa(Xh, Z, C, np.ones(k), dtype=dt, **kw)
            t_hip = time.perf_counter() - t0
        rec_h = 0.5 * np.linalg.norm(X - hZ.dot(hC.dot(X))) ** 2 / n_small
        return {"hip_cost": hcost, "hip_reconstruction_error": rec_h,
                "rel_diff_reconstruction_error": abs(rec_h - rec_o) / rec_o,
                "argmax_equal": bool(np.array_equal(oC.argmax(axis=1), hC.argmax(axis=1))),
                "constraints_ok": bool(np.all(hC >= 0) and np.all(hZ >= 0)
                                       and np.allclose(hC.sum(axis=1), 1, rtol=0, atol=1e-12)
                                       and np.allclose(hZ.sum(axis=1), 1, rtol=0, atol=1e-12)),
                "seconds_hip": t_hip}

    out = {"rows": n_small, "outer_iterations": n_outer, "oracle_cost": ocost,
           "oracle_reconstruction_error": rec_o, "seconds_cpu": t_cpu, "dtype": dtype}
    out.update(leg(dtype))
    if dtype != "float64":
        out["float64"] = leg("float64")
    try:
        fx = np.load(os.path.join(ROOT, "tests", "golden", "converged_1500.npz"))
        if (n_small, n_outer, k) == (1500, 250, 32):
            out["oracle_twin_rel"] = float(fx["twin_rel"])
            out["bound"] = max(1e-5, 20.0 * float(fx["twin_rel"]))
    except (OSError, KeyError):
        pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", type=int, default=N_SAMPLES)
    ap.add_argument("--p", type=int, default=N_FEATURES)
    ap.add_argument("--k", type=int, default=N_COMPONENTS)
    ap.add_argument("--dtype", default="float32", choices=["float32", "float64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # SURVEY.md 8(d) allows a row sample for the CPU baseline; 40 000 rows: from 32 768 rows per GPU on
    # the product runs the pass kernels it runs at full size, so `parity_on_sample` puts the TIMED
    # kernels (k_row_local_f32_dma, k_reduce_rows_f32<1,4>) against the oracle, not their short-shard siblings
    ap.add_argument("--cpu-sample", type=int, default=40000)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--no-f64", action="store_true", help="skip the float64 (reference dtype) leg")
    ap.add_argument("--clock-warmup-ms", type=float, default=300.0,
                    help="run the two pass kernels on scratch operands this long before the W warm-up steps "
                         "(GPU clock ramp after the host-side data generation; 0: off)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs one rank per GPU: launch with "
                             "python -m torch.distributed.run --nproc-per-node %d bench.py ..."
                             % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    from convex_dim_red import _backend
    _backend.require_gpu()                                 # no CPU fallback

    n, p, k = args.n, args.p, args.k
    bounds = np.linspace(0, n, world + 1).astype(np.int64)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    t_gen = time.perf_counter()
    X = synthetic_rows(lo, hi, n, p, k)
    C0, Z0 = start_factors(n, k)
    t_gen = time.perf_counter() - t_gen

    spg_kw = dict(max_iterations=1)                        # production setting (run_hadisst_aa.py:165)
    qp_kw = {}                                             # defaults (archetypal_analysis.py:372-383)

    ctx = _backend.Context(dtype=args.dtype, device=local_rank)
    uid_path = None
    if world > 1:
        if _backend.comm_transport() == "p2p":             # AA_COMM=p2p: one-shot peer-to-peer all-reduce
            ctx.p2p_init(rank, world, "bench")
        else:
            uid, uid_path = exchange_unique_id(rank, world, _backend)
            ctx.comm_init(uid, rank, world)
    ctx.set_data(X, n_global=n, row_offset=lo)
    ctx.set_state(np.ascontiguousarray(C0[:, lo:hi]), Z0[lo:hi], np.ones(k))
    cost0 = ctx.prepare()

    # clock warm-up (not part of the W warm-up steps, touches no solver state): the GPU comes out of the
    # host-side data generation idle; the two pass kernels run on scratch operands for --clock-warmup-ms
    # so that the W + K measured iterations see the clocks a running job sees
    if args.clock_warmup_ms > 0:
        t_w = time.perf_counter()
        while (time.perf_counter() - t_w) * 1e3 < args.clock_warmup_ms:
            ctx.time_kernel(0, 20)
            ctx.time_kernel(1, 20)

    if args.warmup > 0:
        ctx.outer_iterations(args.warmup, spg_kw, qp_kw)
    ctx.allreduce_host([0.0])                              # barrier (outer_iterations ends synchronised)
    t0 = time.perf_counter()
    costs = ctx.outer_iterations(args.steps, spg_kw, qp_kw)
    elapsed = time.perf_counter() - t0                     # host has read the last cost: device idle
    elapsed = float(ctx.allreduce_host([elapsed], "max")[0])

    # the loop the estimators run (aa_iterate: the same updates plus the device-side monotonicity
    # check / stopping rule / conditional snapshot after every iteration, host polling every 8),
    # timed on the same state: tolerance 0 never fires, so exactly `steps` iterations run
    # (single rank only: the device loop has not run on more than one GPU yet, and nothing after
    # the timed region may put a multi-GPU bench line at risk; tools/two_rank_check.py covers it)
    elapsed_loop = None
    if world == 1:
        t0 = time.perf_counter()
        _, st_loop = ctx.iterate(float(costs[-1]), args.steps, 0.0, "abs_delta_f", False, True, True, spg_kw, qp_kw)
        elapsed_loop = time.perf_counter() - t0

    # dominant kernels, timed live with HIP events on the solver's stream: every launch of the
    # two pass kernels inside 10 further outer iterations is bracketed by an event pair
    # (in context, i.e. with the clocks and cache state the timed region has; a back-to-back
    # loop of one kernel runs ~10 % slower)
    ctx.gemm_timing(True)
    ctx.outer_iterations(10, spg_kw, qp_kw)
    ms_reduce, n_reduce, ms_local, n_local = ctx.gemm_timing(False)   # C X / D X / Z'X ; (CX) X' / X (X'Z)
    pass_kernels = ctx.pass_kernels()
    ms_probe = ctx.time_kernel(5, 10)                      # plain streaming read of X (best probe shape)
    qp_stats = ctx.weights_update(**qp_kw)                 # one more QP pass for its statistics
    recon = ctx.reconstruction_cost()
    trace_cost = ctx.cost()
    ms_reduce = float(ctx.allreduce_host([ms_reduce], "max")[0])
    ms_local = float(ctx.allreduce_host([ms_local], "max")[0])
    ctx.close()
    if uid_path and rank == 0:
        try:
            os.remove(uid_path)
        except OSError:
            pass
    if rank != 0:
        return

    # HBM traffic of the two pass kernels from the PMC counters (rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE in separate passes, tools/gpu_pmc.sh; FETCH_SIZE doubled as the MI355X guide
    # prescribes for 16-B/lane streaming reads on gfx950).  Counters cannot be read from
    # inside this process, so the last committed measurement of the same configuration is
    # reported (profiles/pmc_traffic.json) -- null when there is none for this shape.
    traffic = None
    traffic_note = "no PMC measurement committed for this shape / these kernels"
    try:
        import hashlib
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pmc = json.load(fh)
        with open(os.path.join(ROOT, "matrix-factorization-case-studies_amd", "csrc", "kernels_gemm.hip"), "rb") as fh:
            sha = hashlib.sha1(fh.read()).hexdigest()
        # only a measurement of THESE kernels (same source file) on THIS shape is reported
        if (pmc.get("n"), pmc.get("p"), pmc.get("k"), pmc.get("n_gpus")) == (n, p, k, world) \
                and pmc.get("dtype") == args.dtype and pmc.get("kernels_gemm_sha1") == sha:
            traffic = 0.5 * (pmc["reduce_rows_bytes"] + pmc["row_local_bytes"])
            traffic_note = ("profiles/pmc_traffic.json (kernels_gemm.hip sha1 %s): reduce_rows %.4g B, "
                            "row_local %.4g B per launch" % (sha[:12], pmc["reduce_rows_bytes"],
                                                             pmc["row_local_bytes"]))
    except (OSError, ValueError, KeyError):
        pass

    # MFMA utilisation from the SQ counters (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE,
    # tools/gpu_pmc_mfma.sh): busy cycles of the matrix pipes over SIMD-cycles of the dispatch; like
    # the traffic counters it cannot be read from inside this process, so the committed
    # measurement of this configuration is reported next to the flops/time figure
    mfma_pmc = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_mfma.json")) as fh:
            pm = json.load(fh)["kernels"]
        if (n, p, k, world, args.dtype) == (N_SAMPLES, N_FEATURES, N_COMPONENTS, 1, "float32"):
            mfma_pmc = {name.split("<")[0]: round(v["mfma_busy_frac_of_simd_cycles"], 4)
                        for name, v in pm.items() if "mfma_busy_frac_of_simd_cycles" in v}
    except (OSError, ValueError, KeyError):
        pass

    es = 4 if args.dtype == "float32" else 8
    n_loc = hi - lo
    bytes_pass = float(n_loc) * p * es                     # algorithmic bytes of one pass over X
    flops_pass = 2.0 * k * n_loc * p
    ms_dom = 0.5 * (ms_reduce + ms_local)                  # 2 + 2 launches per outer iteration
    achieved_gbs = bytes_pass / (ms_dom * 1e-3) / 1e9
    its = args.steps / elapsed
    flops_alg = 12.0 * k * n * p                           # 6 passes x 2knp
    result = {
        "metric": "AA SPG outer iterations/sec (synthetic %dx%d, k=%d)" % (n, p, k),
        "value": its,
        "unit": "it/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32" if args.dtype == "float32" else "f64",
        "data": "synthetic",
        "config": {"workload": "AA n=%d p=%d k=%d, X row-sharded over %d GPU(s), init=random, "
                               "delta=0, dictionary spg max_iterations=1, default weights QP"
                               % (n, p, k, world),
                   "parallelism": "rows/%d" % world},
        "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                     "kernel": "%s / %s (mean of the two pass kernels, event pairs around every launch in 10 outer iterations)" % pass_kernels,
                     "ms_reduce_rows": ms_reduce, "ms_row_local": ms_local,
                     "launches_timed": [n_reduce, n_local],
                     "ms_streaming_read_probe": ms_probe,
                     "bytes_per_launch": bytes_pass, "flops_per_launch": flops_pass,
                     "mfma_frac_of_kernel": flops_pass / (ms_dom * 1e-3) / (MFMA_F32_PEAK_TFLOPS * 1e12),
                     "mfma_busy_frac_pmc": mfma_pmc,
                     "mfma_busy_source": "profiles/pmc_mfma.json (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), per kernel)"},
        "mfma_frac_outer_iteration": flops_alg / world / (elapsed / args.steps) / (MFMA_F32_PEAK_TFLOPS * 1e12),
        "cost": {"initial": cost0, "final_trace_form": trace_cost, "final_residual_form": recon,
                 "after_each_update_last": [float(costs[-2]), float(costs[-1])]},
        "estimator_loop": None if elapsed_loop is None else {
            "value": args.steps / elapsed_loop, "unit": "it/s",
            "ms_per_step": 1e3 * elapsed_loop / args.steps,
            "what": "aa_iterate, the loop ArchetypalAnalysis.fit_transform runs (device-side "
                    "monotonicity check + stopping rule + snapshot per iteration), %d iterations"
                    % (st_loop.n_iter + 1)},
        "qp": {"mean_passes_per_sample": qp_stats.total_passes / float(n_loc),
               "max_passes": qp_stats.max_passes, "samples_finished_by_wave_kernel": qp_stats.reserved},
        "datagen_s": t_gen,
        # the two pass kernels run on scratch operands for this long BEFORE the W warm-up steps (no solver state
        # touched): the GPU leaves the host-side data generation with idle clocks, and 5 warm-up iterations are
        # 10 ms.  Without it the same window reads ~1 % lower (profiles/round4_ab.txt, call 5f)
        "clock_warmup_ms": args.clock_warmup_ms,
    }

    if not args.no_cpu_baseline and world == 1:
        base = cpu_baseline(k, p, spg_kw, min(args.cpu_sample, n), args.cpu_steps)
        ns = min(args.cpu_sample, n)
        # the same sample through the HIP path for a like-for-like cost comparison
        with _backend.Context(dtype=args.dtype, device=local_rank) as c2:
            c2.set_data(base["X"].astype(np.float32) if args.dtype == "float32" else base["X"])
            c2.set_state(base["C"], base["Z"], np.ones(k))
            c2.prepare()
            gcosts = c2.outer_iterations(1 + args.cpu_steps, spg_kw, qp_kw)
            sample_kernels = c2.pass_kernels()
        result["cpu_baseline"] = {
            "value": base["its_sample"] * ns / float(n),
            "unit": "it/s",
            "cores": len(os.sched_getaffinity(0)),
            "kind": "port",
            "sample": "oracle.iterate_aa (reference op sequence, float64, NumPy/BLAS GEMMs on all "
                      "host cores + %s projection/QP loops) on the first %d rows of the workload, "
                      "%d outer iterations in %.1f s = %.3f it/s at n=%d; value = that rate x %d/%d"
                      % ("serial C" if base["used_c"] else "NumPy", ns, args.cpu_steps,
                         base["seconds"], base["its_sample"], ns, ns, n),
            "phase_seconds": base["timings"],
            "phase_note": "dictionary = spg() on C (7 GEMM passes per SPG iteration + sort-based row "
                          "projections), gram = the four refresh GEMMs, weights = the n per-sample QPs",
        }
        # the same sample through the float64 path (the reference dtype): separates the float32
        # rounding of the two big contractions from everything else
        with _backend.Context(dtype="float64", device=local_rank) as c3:
            c3.set_data(base["X"])
            c3.set_state(base["C"], base["Z"], np.ones(k))
            c3.prepare()
            dcosts = c3.outer_iterations(1 + args.cpu_steps, spg_kw, qp_kw)
        result["parity_on_sample"] = {
            "what": "cost after %d outer iterations from the same start, first %d rows" % (1 + args.cpu_steps, ns),
            "pass_kernels": list(sample_kernels),
            "oracle_cost": base["cost"], "hip_cost": float(gcosts[-1]),
            "rel_diff": abs(float(gcosts[-1]) - base["cost"]) / base["cost"],
            "hip_cost_float64": float(dcosts[-1]),
            "rel_diff_float64": abs(float(dcosts[-1]) - base["cost"]) / base["cost"],
            "note": "float64 leg: the HIP path in the reference dtype against the oracle (algorithmic "
                    "restatements only); the float32 leg adds the rounding of X, of the MFMA operands "
                    "and of the fp32 accumulation",
        }
        result["parity_converged"] = converged_parity(k, p, args.dtype, local_rank, spg_kw, qp_kw)
    if world == 1 and args.dtype == "float32" and not args.no_f64:
        # the reference dtype (the estimators' default) on the same workload: X as float64 (3.3 GB)
        Xd = X.astype(np.float64)
        with _backend.Context(dtype="float64", device=local_rank) as c4:
            c4.set_data(Xd)
            c4.set_state(C0, Z0, np.ones(k))
            c4.prepare()
            c4.outer_iterations(args.warmup, spg_kw, qp_kw)
            n64 = max(10, args.steps // 2)
            t0 = time.perf_counter()
            c64 = c4.outer_iterations(n64, spg_kw, qp_kw)
            t64 = time.perf_counter() - t0
            c4.gemm_timing(True)
            c4.outer_iterations(5, spg_kw, qp_kw)
            r64, nr64, l64, nl64 = c4.gemm_timing(False)
        del Xd
        result["float64"] = {"value": n64 / t64, "unit": "it/s", "ms_per_step": 1e3 * t64 / n64, "steps": n64,
                             "ms_reduce_rows": r64, "ms_row_local": l64,
                             "hbm_frac": float(n) * p * 8 / (0.5 * (r64 + l64) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "cost_last": float(c64[-1])}
    print(json.dumps(result))


if __name__ == "__main__":
    main()
