"""
Sharding of independent Monte-Carlo trajectories across ranks (one process per GPU) and the final aggregation of the
error sums - the only collective of the path.

Trajectories never interact inside the moment-transform / filter path (the reference loops `for imc in range(mc)`,
research/tpq/tpq_base.py:187-189), so each rank owns a contiguous slice of the MC index and no data-path collective
exists.  Phase 1 all-reduces per-time-step sums (squared error, RMSE, NLL, MSE matrix, counts) that a reduction kernel
produced from the filter's own output buffers; phase 2 - only if the
log-credibility ratio is wanted, because it needs the GLOBAL MSE matrix per step (utils.py:113-120 via
research/tpq/tpq_base.py:167-169) - all-reduces the LCR sums.  Messages are a few KB: latency-bound, one fused
all-reduce per phase.  `dist` is `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests) or None for a single process.
"""
import ctypes

import numpy as np

from . import _lib


def shard_bounds(total, rank, world):
    """Contiguous slice [lo, hi) of `total` trajectories owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def device_error_sums(D, B, ld, T, d_x, d_fm, d_fP, d_status=None):
    """Phase-1 sums of this rank's trajectories, reduced on the device (`ssmq_error_sums_dev`).
    d_x, d_fm: DeviceBuffer planes [T][D][ld]; d_fP [T][D*D][ld] (the buffers the filter wrote); d_status [ld] or None.
    Returns a dict of arrays: se (T, D) squared error (utils.py:18-38), rmse (T,) sum of ||x - m|| (the quantity
    research/tpq/tpq_base.py:158-159 averages), nll (T,) negative log-likelihood (utils.py:123-148), mse (T, D, D) outer
    products (utils.py:41-64), n_ok (T,) trajectories counted, n_pd (T,) of them with positive-definite P (nll terms)."""
    lib = _lib.load()
    W = lib.ssmq_error_sums_width(D)
    if W < 0:
        raise _lib.SsmqError('ssmq_error_sums_width: dimension {} out of range'.format(D))
    sums, ps = _lib.out_c((T, W))
    _lib.check(lib.ssmq_error_sums_dev(D, B, ld, T, ctypes.c_void_p(d_x.ptr), ctypes.c_void_p(d_fm.ptr),
                                       ctypes.c_void_p(d_fP.ptr), ctypes.c_void_p(d_status.ptr if d_status else None),
                                       ps), 'ssmq_error_sums_dev')
    return dict(se=sums[:, :D].copy(), rmse=sums[:, D].copy(), nll=sums[:, D + 1].copy(),
                mse=sums[:, D + 2:D + 2 + D * D].reshape(T, D, D).copy(), n_ok=sums[:, D + 2 + D * D].copy(),
                n_pd=sums[:, D + 3 + D * D].copy())


def device_lcr_sums(D, B, ld, T, d_x, d_fm, d_fP, mse_global, d_status=None, reg=1e-6):
    """Phase 2 (`ssmq_lcr_sums_dev`): sums over this rank's trajectories of the log credibility ratio (utils.py:66-120)
    per time step, given the GLOBAL MSE matrices (T, D, D) (+ reg I as research/tpq/tpq_base.py:161-167 does).
    Returns dict lcr (T,), n (T,)."""
    lib = _lib.load()
    M, pM = _lib.as_c(np.asarray(mse_global, dtype=np.float64) + reg * np.eye(D))
    sums, ps = _lib.out_c((T, 2))
    _lib.check(lib.ssmq_lcr_sums_dev(D, B, ld, T, ctypes.c_void_p(d_x.ptr), ctypes.c_void_p(d_fm.ptr),
                                     ctypes.c_void_p(d_fP.ptr), ctypes.c_void_p(d_status.ptr if d_status else None),
                                     pM, ps), 'ssmq_lcr_sums_dev')
    return dict(lcr=sums[:, 0].copy(), n=sums[:, 1].copy())


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
    """Global averages from all-reduced phase-1 sums: rmse_avg (T,), nll_avg (T,), mse (T, D, D), rmse_total (),
    count () trajectories aggregated."""
    n = np.maximum(total['n_ok'], 1.0)
    T = total['rmse'].shape[0]
    return dict(rmse_avg=total['rmse'] / n, nll_avg=total['nll'] / np.maximum(total['n_pd'], 1.0),
                mse=total['mse'] / n[:, None, None], rmse_total=float(np.sqrt(total['se'].sum() / max(total['n_ok'].sum(), 1.0))),
                count=float(total['n_ok'].max()) if T else 0.0)


def finalize_lcr(total):
    """Average log credibility ratio per time step (the inclination indicator's summand) from all-reduced phase-2 sums."""
    return total['lcr'] / np.maximum(total['n'], 1.0)
