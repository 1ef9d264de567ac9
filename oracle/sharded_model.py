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
  projection                       one gathered k x (max, sum, count), then the ranks'
                                   candidate lists in ONE buffer (sum all-reduce of per-rank
                                   slots); fallback: k x (sum, count) per Michelot pass
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


def gather_by_sum(local, comm, rank):
    """What csrc does instead of an all-gather: every rank fills its own slot of a
    [world, ...] buffer (zeros elsewhere) and the buffer is sum-all-reduced."""
    buf = np.zeros((comm.world,) + np.shape(local))
    buf[rank] = local
    return comm.allreduce(buf, "sum").reshape(buf.shape)


def comm_rank(comm):
    return comm.dist.get_rank() if hasattr(comm, "dist") else 0


def project_columns_passes(W, comm, t, max_passes=64):
    """Michelot fixed point from the thresholds t with one all-reduce of k x (sum, count)
    per pass (csrc: k_proj_pass + finalize; also the fallback of the list form)."""
    k = W.shape[1]
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
    return t


def project_columns(W, comm, warm=None, cap=256):
    """Column-wise simplex projection of a tall [n_local, k] array whose rows are sharded,
    as csrc/kernels_tall.hip: launch_proj runs it on several ranks:
      1. per rank (max, sum and count above the warm threshold) -> gathered -> lower bound
         t_lo = max(max - 1, Newton step from the warm threshold)          (k_proj_first)
      2. per rank the candidates {w > t_lo}, at most `cap` per column, gathered in one
         buffer -> every rank runs the fixed point on the union           (k_proj_solve*)
      3. a rank whose list does not fit -> iterative passes from t_lo      (fallback)
    Returns (projection, thresholds)."""
    k = W.shape[1]
    rank = comm_rank(comm)
    tw = np.full(k, np.inf) if warm is None else warm
    above = W > tw
    first = np.stack([W.max(axis=0) if W.shape[0] else np.full(k, -np.inf),
                      np.where(above, W, 0.0).sum(axis=0), above.sum(axis=0).astype(float)])
    allr = gather_by_sum(first, comm, rank)                       # [world, 3, k]
    mx = allr[:, 0].max(axis=0)
    s, cnt = allr[:, 1].sum(axis=0), allr[:, 2].sum(axis=0)
    t = mx - 1.0
    newton = np.where(cnt > 0, (s - 1.0) / np.maximum(cnt, 1.0), -np.inf)
    t = np.where((newton > t) & (newton < mx), newton, t)
    lists = np.full((k, cap + 1), 0.0)
    for i in range(k):
        cand = W[W[:, i] > t[i], i]
        if cand.size <= cap:
            lists[i, :cand.size] = cand
            lists[i, cap] = cand.size
        else:
            lists[i, cap] = -1.0
    allists = gather_by_sum(lists, comm, rank)                    # [world, k, cap + 1]
    if (allists[:, :, cap] < 0).any():
        t = project_columns_passes(W, comm, t)
    else:
        for i in range(k):
            u = np.concatenate([allists[r, i, :int(allists[r, i, cap])] for r in range(comm.world)])
            th, prev = t[i], -1
            for _ in range(200):
                m = u > th
                c = int(m.sum())
                if c == prev or (prev > 0 and c > prev) or c == 0:
                    break
                th = (u[m].sum() - 1.0) / c
                prev = c
            t[i] = th
    return np.fmax(W - t, 0.0), t


def dictionary_update(Xg, Ct, H, ZtZ, alpha, trace, n_global, comm, max_iterations=1,
                      gamma=1e-4, sigma_one=0.1, sigma_two=0.9, lambda_min=1e-10,
                      alpha_min=1e-5, alpha_max=1e3, epsilon_one=1e-10, epsilon_two=1e-6):
    """Data-form dictionary SPG as the device runs it (tall layout: Ct is [n_local, k]).
    Returns (Ct, P, G, f) with P = C X (replicated), G = (C XX')' rows of this shard."""
    k = Ct.shape[1]
    M = alpha[:, None] * ZtZ * alpha[None, :]
    warm = {}
    x, _ = project_columns(Ct, comm)
    P = comm.allreduce(x.T.dot(Xg))
    s1 = comm.allreduce([np.sum(x * H * alpha)])[0]
    a0 = np.trace(M.dot(P.dot(P.T)))
    f_old = 0.5 * (trace - 2 * s1 + a0) / k
    G = Xg.dot(P.T)
    g = (G.dot(M.T) - H * alpha) / n_global
    step = None
    for it in range(max_iterations):
        if step is None:
            pa, warm["alpha"] = project_columns(x - g, comm, warm.get("alpha"))
            reach = comm.allreduce([np.abs(pa - x).max()], "max")[0]
            step = 1.0 / reach if abs(reach) > 1e-12 else 1.0
        pd, warm["dir"] = project_columns(x - step * g, comm, warm.get("dir"))
        d = pd - x
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
        pr, warm["res"] = project_columns(x - g, comm, warm.get("res"))
        res = pr - x
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
