#!/bin/bash
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/../.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -q -s -m gpu --timeout 300 -p no:cacheprovider -x -k "implicit_rbf or kernel_aa" > gpurun_out/r4m_tests.log 2>&1
echo "tests exit=$?"; grep -E "implicit RBF|passed|failed|Error|^E " gpurun_out/r4m_tests.log | tail -20
python - <<'PY' 2>&1 | tail -8
import sys, time, warnings, numpy as np
sys.path.insert(0, "matrix-factorization-case-studies_amd"); sys.path.insert(0, ".")
import convex_dim_red as cdr
warnings.simplefilter("ignore")
rng = np.random.RandomState(0)
for n, p in ((4000, 32), (20000, 100)):
    k = 10
    centers = rng.standard_normal((k, p)) * 2
    X = centers[rng.randint(k, size=n)] + 0.5 * rng.standard_normal((n, p))
    m = cdr.KernelAA(k, init="random", random_state=0, tolerance=0, max_iterations=20, dictionary_solver_kwargs=dict(max_iterations=1), require_monotonic_cost_decrease=False)
    t0 = time.perf_counter(); m.fit_transform(X, features=True, kernel="rbf", gamma=1.0 / p); t = time.perf_counter() - t0
    print("KernelAA implicit RBF n=%d p=%d k=%d: 20 outer iterations in %.3f s (%.1f ms per iteration; the explicit kernel matrix would be %.1f GB), cost %.6f" % (n, p, k, t, 1e3 * t / 20, n * n * 8 / 1e9, m.cost), flush=True)
PY
