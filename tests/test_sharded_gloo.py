"""Multi-rank path on CPU: the device algorithm's row-sharded decomposition
(oracle/sharded_model.py, the NumPy model of csrc/solver.hip) run (a) unsharded and
checked against the reference-pinned oracle, and (b) on 2 ranks over torch.distributed
``gloo``, checked for shard-count invariance."""
import os
import subprocess
import sys
import warnings

import numpy as np

from conftest import ROOT
from oracle import aa_oracle as orc
from oracle import sharded_model as sm


def _problem(seed=3, n=240, p=36, k=5):
    rng = np.random.RandomState(seed)
    B = rng.standard_normal((k, p))
    Zt = orc.right_stochastic_matrix((n, k), rng) ** 3
    Zt /= Zt.sum(axis=1, keepdims=True)
    X = Zt.dot(B) + 0.05 * rng.standard_normal((n, p))
    C0 = orc.right_stochastic_matrix((k, n), rng)
    Z0 = orc.right_stochastic_matrix((n, k), rng)
    return X, C0, Z0


def test_device_model_matches_oracle_unsharded():
    X, C0, Z0 = _problem()
    n, k = Z0.shape
    C, Z, costs = sm.outer_iterations(X, C0, Z0, n, sm.LocalComm(), n_outer=3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wZ, wC, _, wcost, _, _, wdeltas = orc.iterate_aa(
            X, Z0.copy(), C0.copy(), np.ones(k), tolerance=0, max_iterations=3,
            dictionary_solver_kwargs=dict(max_iterations=1), require_monotonic_cost_decrease=False)
    # three outer iterations: the per-sample QPs stop at their 1e-6 tolerance, so last-bit
    # differences of a projection move the cost at the 1e-8 level
    assert abs(costs[-1, 1] - wcost) < 1e-7 * wcost
    assert np.abs(C - wC).max() < 1e-7
    assert np.abs(Z - wZ).max() < 1e-5
    assert np.array_equal(C > 0, wC > 0)


def test_model_list_projection_matches_sorted_scan():
    """The device's list form of the column projection (lower bound from one Newton step,
    candidate lists gathered across ranks, fixed point on the union; iterative passes when a
    list overflows) against the reference's sorted scan, cold, warm and overflowing."""
    rng = np.random.RandomState(3)
    W = 0.3 * rng.standard_normal((700, 5)) + 0.05
    want = np.stack([orc.simplex_project_vector_py(W[:, i]) for i in range(5)], axis=1)
    comm = sm.LocalComm()
    P, t = sm.project_columns(W, comm)
    assert np.abs(P - want).max() < 1e-15
    W2 = W + 0.02 * rng.standard_normal(W.shape)
    want2 = np.stack([orc.simplex_project_vector_py(W2[:, i]) for i in range(5)], axis=1)
    P2, _ = sm.project_columns(W2, comm, warm=t)             # warm thresholds of a nearby problem
    assert np.abs(P2 - want2).max() < 1e-15
    P3, _ = sm.project_columns(W2, comm, warm=t + 5.0)       # useless warm start (empty support)
    assert np.abs(P3 - want2).max() < 1e-15
    P4, _ = sm.project_columns(W, comm, cap=2)               # every list overflows -> passes
    assert np.abs(P4 - want).max() < 1e-15


def test_device_model_multi_iteration_spg_matches_oracle():
    X, C0, Z0 = _problem(seed=8)
    n, k = Z0.shape
    trace = np.sum(X * X)
    ZtZ = Z0.T.dot(Z0)
    H = X.dot(X.T.dot(Z0))
    Ct, P, G, f = sm.dictionary_update(X, np.ascontiguousarray(C0.T), H, ZtZ, np.ones(k), trace, n,
                                       sm.LocalComm(), max_iterations=5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wC, wf, _, _ = orc.update_aa_dictionary(X, C0, np.ones(k), trace, H, ZtZ, max_iterations=5)
    assert np.abs(Ct.T - wC).max() < 1e-9
    assert abs(f - wf) < 1e-11 * abs(wf)


_WORKER = r"""
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
from oracle import sharded_model as sm
from test_sharded_gloo import _problem
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
X, C0, Z0 = _problem()
n = X.shape[0]
bounds = np.linspace(0, n, world + 1).astype(int)
lo, hi = bounds[rank], bounds[rank + 1]
C, Z, costs = sm.outer_iterations(X[lo:hi], C0[:, lo:hi], Z0[lo:hi], n, sm.TorchComm(), n_outer=3)
np.savez(os.path.join(%(out)r, "rank%%d.npz" %% rank), C=C, Z=Z, costs=costs, lo=lo, hi=hi)
dist.barrier()
dist.destroy_process_group()
"""


def test_two_rank_gloo_shard_invariance(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % dict(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    X, C0, Z0 = _problem()
    n = X.shape[0]
    wC, wZ, wcosts = sm.outer_iterations(X, C0, Z0, n, sm.LocalComm(), n_outer=3)
    for rank in range(2):
        r = np.load(tmp_path / ("rank%d.npz" % rank))
        lo, hi = int(r["lo"]), int(r["hi"])
        # first two iterations agree to rounding; later ones to the QP's own stopping
        # tolerance (1e-6 on the weights), which feeds back into the next iterate
        assert np.abs(r["costs"][:2] - wcosts[:2]).max() < 1e-11 * abs(wcosts).max()
        assert np.abs(r["costs"] - wcosts).max() < 1e-7 * abs(wcosts).max()
        assert np.abs(r["C"] - wC[:, lo:hi]).max() < 1e-6
        assert np.abs(r["Z"] - wZ[lo:hi]).max() < 1e-4
        assert np.array_equal(r["C"] > 0, wC[:, lo:hi] > 0)
