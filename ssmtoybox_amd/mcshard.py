"""
Sharding of independent Monte-Carlo trajectories across ranks (one process per GPU) and the final aggregation of the
error sums - the only collective of the path.

Trajectories never interact inside the moment-transform / filter path (the reference loops `for imc in range(mc)`,
research/tpq/tpq_base.py:187-189), so each rank owns a contiguous slice of the MC index and no data-path collective
exists.  Phase 1 all-reduces per-time-step sums (squared error, RMSE, NLL, MSE matrix, count); phase 2 - only if the
log-credibility ratio is wanted, because it needs the GLOBAL MSE matrix per step (utils.py:113-120 via
research/tpq/tpq_base.py:167-169) - all-reduces the LCR sums.  Messages are a few KB: latency-bound, one fused
all-reduce per phase.  `dist` is `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests) or None for a single process.
"""
import numpy as np


def shard_bounds(total, rank, world):
    """Contiguous slice [lo, hi) of `total` trajectories owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def local_error_sums(x_true, fm, fP, ok=None):
    """Per-time-step sums over this rank's trajectories.
    x_true, fm: (D, T, B); fP: (D, D, T, B); ok: (B,) bool mask of trajectories that did not fail.
    Returns a dict of arrays: se (T, D) squared error (utils.py:18-38), rmse (T,) sum of ||x - m|| (the quantity
    research/tpq/tpq_base.py:158-159 averages), nll (T,) negative log-likelihood (utils.py:123-148), mse (T, D, D) outer
    products (utils.py:41-64), count ()."""
    D, T, B = fm.shape
    ok = np.ones(B, dtype=bool) if ok is None else np.asarray(ok, dtype=bool)
    dx = (x_true - fm)[:, :, ok]                                   # (D, T, b)
    P = fP[:, :, :, ok].transpose(2, 3, 0, 1)                      # (T, b, D, D)
    se = (dx ** 2).sum(axis=2).T
    rmse = np.sqrt((dx ** 2).sum(axis=0)).sum(axis=1)
    mse = np.einsum('itb,jtb->tij', dx, dx)
    d = dx.transpose(1, 2, 0)                                      # (T, b, D)
    sol = np.linalg.solve(P, d[..., None])[..., 0]
    sign, logdet = np.linalg.slogdet(P)
    nll = (0.5 * (sign * logdet + (d * sol).sum(axis=-1) + D * np.log(2 * np.pi))).sum(axis=1)
    return dict(se=se, rmse=rmse, nll=nll, mse=mse, count=np.array(float(ok.sum())))


def _pack(sums, keys):
    return np.concatenate([np.asarray(sums[k], dtype=np.float64).reshape(-1) for k in keys])


def _unpack(flat, sums, keys):
    out, pos = {}, 0
    for k in keys:
        n = int(np.asarray(sums[k]).size)
        out[k] = flat[pos:pos + n].reshape(np.asarray(sums[k]).shape)
        pos += n
    return out


def allreduce_sums(sums, dist=None, device=None):
    """Sum a dict of arrays over all ranks with ONE all-reduce of the packed buffer."""
    keys = sorted(sums)
    flat = _pack(sums, keys)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        import torch
        if device is None:
            device = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
        t = torch.from_numpy(flat.copy()).to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        flat = t.cpu().numpy()
    return _unpack(flat, sums, keys)


def finalize(total):
    """Global averages from all-reduced sums: rmse_avg (T,), nll_avg (T,), mse (T, D, D), rmse_total ()."""
    n = max(float(total['count']), 1.0)
    T = total['rmse'].shape[0]
    return dict(rmse_avg=total['rmse'] / n, nll_avg=total['nll'] / n, mse=total['mse'] / n,
                rmse_total=float(np.sqrt(total['se'].sum() / (n * T))), count=n)


def local_lcr_sums(x_true, fm, fP, mse_global, ok=None, reg=1e-6):
    """Phase 2: sums over this rank's trajectories of the log credibility ratio (utils.py:66-120) per time step, given the
    global MSE matrices (+ reg I as research/tpq/tpq_base.py:161-167 does).  Returns dict(lcr (T,))."""
    D, T, B = fm.shape
    ok = np.ones(B, dtype=bool) if ok is None else np.asarray(ok, dtype=bool)
    dx = (x_true - fm)[:, :, ok].transpose(1, 2, 0)                # (T, b, D)
    P = fP[:, :, :, ok].transpose(2, 3, 0, 1)
    M = mse_global + reg * np.eye(D)
    a = (dx * np.linalg.solve(P, dx[..., None])[..., 0]).sum(axis=-1)
    b = (dx * np.linalg.solve(M[:, None], dx[..., None])[..., 0]).sum(axis=-1)
    return dict(lcr=(10 * (np.log10(a) - np.log10(b))).sum(axis=1))
