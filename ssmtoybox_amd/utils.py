"""
The helper functions of the reference's `ssmtoybox/utils.py` that lie on the accelerated path, under their own names:
the multi-index / Vandermonde helpers of the Bayes-Sard weights (utils.py:459-502) and the performance metrics
(utils.py:41-148).  Everything numerical runs on the device through the C ABI; there is no NumPy fallback.

The metric functions keep the reference's PER-ITEM signatures (one state, one mean, one covariance) and are a thin
convenience: a Monte-Carlo study should reduce its filter outputs where they lie with `mcshard.device_error_sums` /
`device_lcr_sums` (one launch for all trajectories and steps) instead of calling these in a loop, which is what
research/tpq/tpq_base.py:154-172 does on the CPU.  `squared_error` is not restated: its aggregate over the Monte-Carlo
axis is the `se` entry of `mcshard.device_error_sums`.
"""

import numpy as np

from . import _lib, mcshard
from .bq.bqmod import n_sum_k  # noqa: F401  (utils.py:459-475; integer code, defined next to its only user)


def vandermonde(mul_ind, x):
    """utils.py:478-502: (num_points, num_basis) matrix of the monomials x_n ** mul_ind[:, b] (`ssmq_bs_moments`)."""
    mi = np.ascontiguousarray(mul_ind, dtype=np.int32)
    xs, px = _lib.as_c(np.atleast_2d(np.asarray(x, dtype=np.float64)))
    D, N = xs.shape
    if mi.ndim != 2 or mi.shape[0] != D:
        raise ValueError('multi-indices must have one row per dimension of the points')
    out, pout = _lib.out_c((N, mi.shape[1]))
    _lib.check(_lib.load().ssmq_bs_moments(D, N, px, None, mi.ctypes.data_as(_lib.c_int32_p), mi.shape[1], None, None,
                                           None, None, pout), 'ssmq_bs_moments')
    return out


def _planes(a, ld):
    """(D, B) or (D, D, B) host array -> planes [D..][ld] in HBM (one time step)."""
    src = np.asarray(a, dtype=np.float64).reshape(-1, a.shape[-1])
    buf = np.zeros((src.shape[0], ld))
    buf[:, :src.shape[1]] = src
    d = _lib.DeviceBuffer(buf.nbytes)
    d.upload(buf)
    return d


def _one_step(x, m, P):
    """Upload B states / means / covariances of ONE time step; returns (D, B, ld, d_x, d_m, d_P)."""
    x, m = np.asarray(x, dtype=np.float64), np.asarray(m, dtype=np.float64)
    D = m.shape[0]
    m2 = m.reshape(D, -1)
    B = m2.shape[1]
    x2 = np.broadcast_to(x.reshape(D, -1), (D, B))
    ld = max(64, (B + 63) // 64 * 64)
    if P is None:
        P = np.broadcast_to(np.eye(D)[:, :, None], (D, D, B))
    P3 = np.asarray(P, dtype=np.float64).reshape(D, D, -1)
    return D, B, ld, _planes(x2, ld), _planes(m2, ld), _planes(np.broadcast_to(P3, (D, D, B)), ld)


def mse_matrix(x, m):
    """utils.py:41-64: sample mean-square-error matrix of the estimates m (dim, mc) of the state(s) x (dim, 1 | mc)."""
    D, B, ld, d_x, d_m, d_P = _one_step(x, m, None)
    try:
        s = mcshard.device_error_sums(D, B, ld, 1, d_x, d_m, d_P)
    finally:
        for b in (d_x, d_m, d_P):
            b.free()
    return s['mse'][0] / B


def neg_log_likelihood(x, m, P):
    """utils.py:123-148: 0.5 (sign log|det P| + dx' inv(P) dx + d log 2 pi) of one estimate; raises LinAlgError for a
    singular P, where numpy.linalg.inv raises in the reference."""
    D, B, ld, d_x, d_m, d_P = _one_step(x, m, P)
    try:
        s = mcshard.device_error_sums(D, B, ld, 1, d_x, d_m, d_P)
    finally:
        for b in (d_x, d_m, d_P):
            b.free()
    if s['n_pd'][0] != B:
        raise np.linalg.LinAlgError('Singular matrix')
    return float(s['nll'][0])


def log_cred_ratio(x, m, P, MSE):
    """utils.py:66-120: 10 (log10 dx' P^-1 dx - log10 dx' MSE^-1 dx) of one estimate, with `mat_sqrt`'s SVD route for a
    P that is not positive definite (utils.py:412-433).  MSE must be symmetric positive definite here (a sample MSE
    matrix plus its regulariser, research/tpq/tpq_base.py:161-167)."""
    D, B, ld, d_x, d_m, d_P = _one_step(x, m, P)
    try:
        s = mcshard.device_lcr_sums(D, B, ld, 1, d_x, d_m, d_P, np.asarray(MSE, dtype=np.float64).reshape(1, D, D),
                                    reg=0.0)
    finally:
        for b in (d_x, d_m, d_P):
            b.free()
    if s['n'][0] != B:
        raise np.linalg.LinAlgError('log_cred_ratio: covariance or MSE matrix is singular')
    return float(s['lcr'][0])
