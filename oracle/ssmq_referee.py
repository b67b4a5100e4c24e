"""
Extended-precision referee (TEST INFRASTRUCTURE - never imported by the package, never timed).

Some recursions of the path are compared with the reference only at the level of the reference's own rounding noise:
the BQ covariance is the UNCENTRED difference fx Wc fx' - m m' (bq/bqmtran.py:199), which on the reentry model cancels
4e7-sized terms into 1e-6-sized variances, so two correct fp64 evaluations of the same filter differ by 1e-2 RELATIVE
in P from the first step on.  Three fp64 results that disagree by a few per cent pin nothing by themselves.  This module
restates the same recursions in 40-digit arithmetic (mpmath) from the SAME fp64 inputs (weights, sigma points,
measurements): the exact result of the algorithm.  A test can then demand
        |device - exact|  <=  c |reference - exact|
i.e. that the device is no further from the algorithm's exact value than the reference's own NumPy evaluation is.

Follows, line by line like oracle/ssmq_oracle.py: BQTransform.apply bq/bqmtran.py:60-109, _mean/_covariance/
_cross_covariance :158-223, the t-process model variance :394-415 + bq/bqmod.py:1132-1160, the integrands ssmod.py:530-564
(reentry), :1227-1252 (radar), :675-690 (coordinated turn), :1189-1195 (bearings), the Gaussian recursion
ssinf.py:66-118, 254-323.
"""
import mpmath as mp
import numpy as np

DPS = 40

F_REENTRY2D_DYN, F_RADAR2D_MEAS, F_CT_DYN, F_BEARING_MEAS = 9, 10, 11, 12
F_UNGM_DYN, F_UNGM_MEAS = 1, 2


def _m(a):
    a = np.asarray(a, dtype=float)
    if a.ndim == 1:
        return mp.matrix([mp.mpf(float(v)) for v in a])
    return mp.matrix([[mp.mpf(float(v)) for v in row] for row in a])


def _np(m):
    return np.array([[float(m[i, j]) for j in range(m.cols)] for i in range(m.rows)])


def integrand(fid, x, t, p=()):
    """x: list of mpf (one sigma point); returns a list of mpf.  Formulas of ssmod.py as oracle.integrand states them."""
    if fid == F_UNGM_DYN:
        return [mp.mpf(0.5) * x[0] + 25 * (x[0] / (1 + x[0] ** 2)) + 8 * mp.cos(mp.mpf(1.2) * t)]
    if fid == F_UNGM_MEAS:
        return [mp.mpf(0.05) * x[0] ** 2]
    if fid == F_REENTRY2D_DYN:
        dt = mp.mpf(float(p[0]))
        r0, h0, gm0, b0 = mp.mpf(6374.0), mp.mpf(13.406), mp.mpf(3.9860e5), mp.mpf(-0.59783)
        b = b0 * mp.exp(x[4])
        rr = mp.sqrt(x[0] ** 2 + x[1] ** 2)
        vv = mp.sqrt(x[2] ** 2 + x[3] ** 2)
        dr = b * mp.exp((r0 - rr) / h0) * vv
        gr = -gm0 / rr ** 3
        return [x[0] + dt * x[2], x[1] + dt * x[3], x[2] + dt * (dr * x[2] + gr * x[0]),
                x[3] + dt * (dr * x[3] + gr * x[1]), x[4]]
    if fid == F_RADAR2D_MEAS:
        lx, ly = (mp.mpf(float(p[0])), mp.mpf(float(p[1]))) if len(p) >= 2 else (mp.mpf(0), mp.mpf(0))
        return [mp.sqrt((x[0] - lx) ** 2 + (x[1] - ly) ** 2), mp.atan2(x[1] - ly, x[0] - lx)]
    if fid == F_CT_DYN:
        dt = mp.mpf(float(p[0]))
        om = x[4]
        a, b = mp.sin(om * dt), mp.cos(om * dt)
        c, d = a / om, (1 - b) / om
        return [x[0] + c * x[1] - d * x[3], b * x[1] - a * x[3], d * x[1] + x[2] + c * x[3], a * x[1] + b * x[3], x[4]]
    if fid == F_BEARING_MEAS:
        sp = np.asarray(p, dtype=float).reshape(-1, 2)
        return [mp.atan2(x[1] - mp.mpf(float(s[1])), x[0] - mp.mpf(float(s[0]))) for s in sp]
    raise ValueError(fid)


def apply_bq(fid, mean, cov, t, pts, w, p=(), state_index=None, tp_nu=None, emv_broadcast=False):
    """One BQ moment transform in extended precision; mean / cov are mp matrices, pts and the weights fp64 arrays (taken
    as exact).  w: dict wm, Wc, Wcc, model_var (scalar or (E, E)), iK for the t-process.  Returns mp matrices."""
    D, N = pts.shape
    L = mp.cholesky(cov)
    xi = _m(pts)
    x = L * xi
    cols = []
    for n in range(N):
        xn = [mean[d] + x[d, n] for d in range(D)]
        if state_index is not None:
            xn = [xn[i] for i in state_index]
        cols.append(integrand(fid, xn, t, p))
    E = len(cols[0])
    fx = mp.matrix(E, N)
    for n in range(N):
        for e in range(E):
            fx[e, n] = cols[n][e]
    wm, Wc, Wcc = _m(w['wm']), _m(w['Wc']), _m(w['Wcc'])
    mean_f = fx * wm
    mv = np.asarray(w['model_var'], dtype=float)
    emv = mp.matrix(E, E)
    if mv.ndim == 2:
        for e in range(E):
            for e2 in range(E):
                if emv_broadcast or e == e2:
                    emv[e, e2] = mp.mpf(float(mv[e, e2] if mv.shape[0] == E else mv[0, 0]))
    else:
        for e in range(E):
            for e2 in range(E):
                if emv_broadcast or e == e2:
                    emv[e, e2] = mp.mpf(float(mv))
    if tp_nu is not None:
        S = fx * _m(w['iK']) * fx.T
        nu = mp.mpf(float(tp_nu))
        for e in range(E):
            for e2 in range(E):
                emv[e, e2] = (nu - 2 + S[e, e2]) / (nu - 2 + N) * emv[e, e2]
    cov_f = fx * Wc * fx.T - mean_f * mean_f.T + emv
    cov_fx = fx * Wcc.T * L.T
    return mean_f, cov_f, cov_fx


def kalman_update(m_pr, P_pr, y_mean, P_y, P_yx, y):
    """ssinf.py:297-323: K = (P_y^-1 P_yx)', m = m- + K (y - y_mean), P = P- - K P_y K' (unsymmetrised)."""
    gain = (mp.inverse(P_y) * P_yx).T          # 40 digits: the explicit inverse loses nothing that matters here
    return m_pr + gain * (y - y_mean), P_pr - gain * P_y * gain.T


def gaussian_filter(y, m0, P0, GQG, Rn, tf_dyn, tf_obs, steps=None):
    """Forward pass (ssinf.py:66-118, 254-323) carried in extended precision; y (dim_y, T) fp64.
    tf_dyn / tf_obs: callables (mean, cov, t) -> mp matrices.  Returns fm (D, T), fP (D, D, T) rounded to fp64."""
    mp.mp.dps = DPS
    D = len(m0)
    T = y.shape[1] if steps is None else steps
    fm, fP = np.zeros((D, T)), np.zeros((D, D, T))
    m, P = _m(m0), _m(P0)
    GQGm, Rm = _m(GQG), _m(Rn)
    for k in range(1, T + 1):
        m_pr, P_pr, _ = tf_dyn(m, P, k - 1)
        P_pr = P_pr + GQGm
        y_mean, P_y, P_yx = tf_obs(m_pr, P_pr, k - 1)
        P_y = P_y + Rm
        m, P = kalman_update(m_pr, P_pr, y_mean, P_y, P_yx, _m(y[:, k - 1]))
        fm[:, k - 1] = [float(v) for v in m]
        fP[..., k - 1] = _np(P)
    return fm, fP


def bq_filter(y, m0, P0, GQG, Rn, fid_dyn, fid_obs, pts, w_dyn, w_obs, p_dyn=(), p_obs=(), idx_obs=None, tp_nu=None,
              emv_broadcast=False, steps=None):
    """A BQ Kalman filter (GPQ / Bayes-Sard / t-process transforms given by their weights) on one measurement sequence."""
    mp.mp.dps = DPS
    tf_dyn = lambda m, P, t: apply_bq(fid_dyn, m, P, t, pts, w_dyn, p_dyn, None, tp_nu, emv_broadcast)
    tf_obs = lambda m, P, t: apply_bq(fid_obs, m, P, t, pts, w_obs, p_obs, idx_obs, tp_nu, emv_broadcast)
    return gaussian_filter(y, m0, P0, GQG, Rn, tf_dyn, tf_obs, steps)
