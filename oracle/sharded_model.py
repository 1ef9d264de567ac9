"""NumPy model of the DEVICE algorithm with the data matrix row-sharded over ranks.

TEST INFRASTRUCTURE (CPU only).  It restates, in NumPy, exactly the decomposition
csrc/solver.hip runs on the GPUs -- which quantities are computed per shard and which
small ones are all-reduced -- so that the multi-GPU path can be rehearsed without GPUs
(world_size-2 ``gloo`` test in tests/test_sharded_gloo.py) and checked for shard-count
invariance against the unsharded oracle (oracle/aa_oracle.py: iterate_aa).

Per outer iteration the collectives are (all float64, sum unless noted):
  P = C X, Q = D X, Z'X            k x p
  Z'Z, C XX'Z, Grams of CX / DX    k x k           (the latter from replicated operands)
  packed scalars                   <d,g>, <d,d>, tr(D H), ||res||^2, (max) |res|_inf, alpha^-1
  projection passes                k x (sum, count) per Michelot pass, k maxima once
Everything else (C XX' rows, X X'Z rows, the n/G per-sample QPs, the row-local part of
every projection) needs no communication.
"""
import numpy as np

from . import aa_oracle as orc


class LocalComm(object):
    """world_size 1."""
    world = 1

    def allreduce(self, a, op="sum"):
        return np.array(a, dtype=np.float64, copy=True)


class TorchComm(object):
    """torch.distributed (gloo on CPU) as the all-reduce."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.world = dist.get_world_size()

    def allreduce(self, a, op="sum"):
        import torch
        t = torch.from_numpy(np.array(a, dtype=np.float64, copy=True))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM)
        return t.numpy()


def project_columns(W, comm, max_passes=64):
    """Column-wise simplex projection of a tall [n_local, k] array whose rows are
    sharded: Michelot fixed point with one all-reduce of k x (sum, count) per pass."""
    k = W.shape[1]
    local_max = W.max(axis=0) if W.shape[0] else np.full(k, -np.inf)
    t = comm.allreduce(local_max, "max") - 1.0
    prev = np.zeros(k)
    done = np.zeros(k, dtype=bool)
    for _ in range(max_passes):
        mask = W > t
        sc = np.concatenate([np.where(mask, W, 0.0).sum(axis=0), mask.sum(axis=0).astype(float)])
        sc = comm.allreduce(sc, "sum")
        s, cnt = sc[:k], sc[k:]
        conv = (prev > 0) & (cnt >= prev)
        t = np.where(done, t, (s - 1.0) / np.maximum(cnt, 1.0))
        prev = np.where(done, prev, cnt)
        done |= conv
        if done.all():
            break
    return np.fmax(W - t, 0.0)


def dictionary_update(Xg, Ct, H, ZtZ, alpha, trace, n_global, comm, max_iterations=1,
                      gamma=1e-4, sigma_one=0.1, sigma_two=0.9, lambda_min=1e-10,
                      alpha_min=1e-5, alpha_max=1e3, epsilon_one=1e-10, epsilon_two=1e-6):
    """Data-form dictionary SPG as the device runs it (tall layout: Ct is [n_local, k]).
    Returns (Ct, P, G, f) with P = C X (replicated), G = (C XX')' rows of this shard."""
    k = Ct.shape[1]
    M = alpha[:, None] * ZtZ * alpha[None, :]
    x = project_columns(Ct, comm)
    P = comm.allreduce(x.T.dot(Xg))
    s1 = comm.allreduce([np.sum(x * H * alpha)])[0]
    a0 = np.trace(M.dot(P.dot(P.T)))
    f_old = 0.5 * (trace - 2 * s1 + a0) / k
    G = Xg.dot(P.T)
    g = (G.dot(M.T) - H * alpha) / n_global
    step = None
    for it in range(max_iterations):
        if step is None:
            reach = comm.allreduce([np.abs(project_columns(x - g, comm) - x).max()], "max")[0]
            step = 1.0 / reach if abs(reach) > 1e-12 else 1.0
        d = project_columns(x - step * g, comm) - x
        delta, dd, s1d = comm.allreduce([np.sum(d * g), np.sum(d * d), np.sum(d * H * alpha)])
        Q = comm.allreduce(d.T.dot(Xg))
        a1 = np.trace(M.dot(P.dot(Q.T))) + np.trace(M.dot(Q.dot(P.T)))
        a2 = np.trace(M.dot(Q.dot(Q.T)))
        f_of = lambda lam: 0.5 * (trace - 2 * (s1 + lam * s1d) + a0 + lam * a1 + lam * lam * a2) / k
        lam = 1.0
        f_new = f_of(lam)
        while f_new > f_old + gamma * lam * delta:
            lam = orc.line_search_step_length(lam, delta, f_old, f_new, sigma_one, sigma_two)
            f_new = f_of(lam)
            if abs(lam) < lambda_min:
                break
        x = x + lam * d
        P = P + lam * Q
        s1 += lam * s1d
        a0 += lam * a1 + lam * lam * a2
        G = Xg.dot(P.T)
        g_new = (G.dot(M.T) - H * alpha) / n_global
        dgn = comm.allreduce([np.sum(d * g_new)])[0]
        step = orc.cauchy_step_size(lam * (dgn - delta), lam * lam * dd, alpha_min, alpha_max)
        f_old = f_new
        g = g_new
        res = project_columns(x - g, comm) - x
        r2 = comm.allreduce([np.sum(res * res)])[0]
        rinf = comm.allreduce([np.abs(res).max() if res.size else 0.0], "max")[0]
        if np.sqrt(r2) < epsilon_two or rinf < epsilon_one:
            break
    return x, P, G, f_old


def outer_iterations(Xg, Cg, Zg, n_global, comm, n_outer=2, spg_kw=None, qp_kw=None):
    """``Cg`` is this rank's COLUMN block of the dictionary (k x n_local), ``Zg`` its row
    block of the weights.  Returns (Cg, Zg, costs[n_outer, 2])."""
    spg_kw = spg_kw or dict(max_iterations=1)
    qp_kw = qp_kw or {}
    k = Cg.shape[0]
    alpha = np.ones(k)
    Ct = np.ascontiguousarray(Cg.T)
    Z = Zg.copy()
    trace = comm.allreduce([np.sum(Xg * Xg)])[0]
    ZtZ = comm.allreduce(Z.T.dot(Z))
    ZtX = comm.allreduce(Z.T.dot(Xg))
    H = Xg.dot(ZtX.T)                                   # XX'Z rows of this shard
    costs = np.zeros((n_outer, 2))

    def cost(P, CKZ, ZtZ):
        return 0.5 * (trace - 2 * np.trace(CKZ) + np.trace(ZtZ.dot(P.dot(P.T)))) / n_global

    for it in range(n_outer):
        Ct, P, G, _ = dictionary_update(Xg, Ct, H, ZtZ, alpha, trace, n_global, comm, **spg_kw)
        CKZ = comm.allreduce(Ct.T.dot(H))
        costs[it, 0] = cost(P, CKZ, ZtZ)
        Z = orc.qp_batch(P.dot(P.T), G, Z, "nk", **qp_kw)       # b_t = -G[t]
        ZtZ = comm.allreduce(Z.T.dot(Z))
        ZtX = comm.allreduce(Z.T.dot(Xg))
        H = Xg.dot(ZtX.T)
        CKZ = comm.allreduce(Ct.T.dot(H))
        costs[it, 1] = cost(P, CKZ, ZtZ)
    return np.ascontiguousarray(Ct.T), Z, costs
