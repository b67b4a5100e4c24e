"""Shared by the CPU and GPU referee tests: the exact (40-digit) results of the two recursions that are otherwise compared
at the reference's own noise level (oracle/ssmq_referee.py), on the inputs of tests/golden/g10_referee.npz, and the
per-step error statistics against them."""
import functools
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g10_referee.npz')


def load():
    return np.load(GOLDEN)


def weights(g, tag):
    w = dict(wm=g[tag + '_wm'], Wc=g[tag + '_Wc'], Wcc=g[tag + '_Wcc'], model_var=g[tag + '_mv'])
    if tag + '_iK' in g.files:
        w['iK'] = g[tag + '_iK']
    return w


@functools.lru_cache(maxsize=None)
def reentry_exact():
    """Bayes-Sard Kalman filter, reentry 5-D + radar (BASELINE configs[2] as the reference's own study configures it):
    exact filtered moments (5, T, S), (5, 5, T, S) for the S measurement sequences of the fixture."""
    from oracle import ssmq_referee as rf
    g = load()
    wd, wo = weights(g, 'rer_dyn'), weights(g, 'rer_obs')
    G = g['rer_G']
    GQG = G.dot(g['rer_Q']).dot(G.T)
    y = g['rer_y']
    fm, fP = np.zeros((5,) + y.shape[1:]), np.zeros((5, 5) + y.shape[1:])
    for s in range(y.shape[2]):
        fm[..., s], fP[..., s] = rf.bq_filter(y[..., s], g['rer_m0'], g['rer_P0'], GQG, g['rer_R'], rf.F_REENTRY2D_DYN,
                                               rf.F_RADAR2D_MEAS, g['rer_dyn_pts'], wd, wo, (0.1,), (0.0, 0.0))
    return fm, fP


@functools.lru_cache(maxsize=None)
def ct_exact():
    """t-process Kalman filter, coordinated turn 5-D + four bearing sensors (BASELINE configs[3]); StudentProcessKalman builds
    its transforms with dim_out = 1, so the model variance is broadcast over the output covariance (ssinf.py:503-553)."""
    from oracle import ssmq_referee as rf
    g = load()
    wd, wo = weights(g, 'ct_dyn'), weights(g, 'ct_obs')
    y = g['ct_y']
    fm, fP = np.zeros((5,) + y.shape[1:]), np.zeros((5, 5) + y.shape[1:])
    p_obs = tuple(g['ct_sensors'].reshape(-1))
    for s in range(y.shape[2]):
        fm[..., s], fP[..., s] = rf.bq_filter(y[..., s], g['ct_m0'], g['ct_P0'], g['ct_Q'], g['ct_R'], rf.F_CT_DYN,
                                               rf.F_BEARING_MEAS, g['ct_dyn_pts'], wd, wo, (float(g['ct_dt'][0]),), p_obs,
                                               idx_obs=(0, 2), tp_nu=float(g['ct_nu'][0]), emv_broadcast=True)
    return fm, fP


def step_errors(fm, fP, xm, xP):
    """Per time step, pooled over trajectories: (max, rms) of the mean error in posterior standard deviations
    |dm_i| / sqrt(P_ii) and of the covariance error |dP_ij| / sqrt(P_ii P_jj), both against the EXACT moments xm, xP.
    Layouts (D, T, S) / (D, D, T, S); trajectories with NaN (failed runs) are left out of both."""
    ok = np.isfinite(fm).all(axis=(0, 1)) & np.isfinite(fP).all(axis=(0, 1, 2))
    sd = np.sqrt(np.abs(np.einsum('iits->its', xP)))
    em = (np.abs(fm - xm) / sd)[:, :, ok]
    eP = (np.abs(fP - xP) / (sd[:, None] * sd[None]))[:, :, :, ok]
    return dict(m_max=em.max(axis=(0, 2)), m_rms=np.sqrt((em ** 2).mean(axis=(0, 2))),
                P_max=eP.max(axis=(0, 1, 3)), P_rms=np.sqrt((eP ** 2).mean(axis=(0, 1, 3))), ok=ok)
