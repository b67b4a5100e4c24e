"""Pin the C restatement of the hot loop (oracle/ssmq_oracle.c) to the reference's golden vectors and to the NumPy
oracle.  CPU only."""
import numpy as np
import pytest

from oracle import ssmq_oracle as orc
from oracle import c_oracle as co
from tests._cases import MODELS, SIGMA_TF, BQ_TF, assert_moments_close, rel_err, cov_err, mean_err


def _c_transform(g, name, tname, din, dout, fid, p, sidx):
    integ = co.Integrand.make(fid, p, sidx)
    key = '{}_{}'.format(name, tname)
    if tname in SIGMA_TF:
        if tname == 'ut':
            pts, (wm, wc) = orc.points_ut(din), orc.weights_ut(din)
        elif tname == 'sr':
            pts, wm, wc = orc.points_sr(din), orc.weights_sr(din), orc.weights_sr(din)
        elif tname == 'gh':
            pts, wm, wc = orc.points_gh(din, 3), orc.weights_gh(din, 3), orc.weights_gh(din, 3)
        else:
            pts, wm, wc = orc.points_fs(din, 3), orc.weights_fs(din, 3), orc.weights_fs(din, 3)
        return co.make_transform(1, din, dout, pts, wm, wc, integrand=integ)
    emv = float(g[key + '_mv']) * np.ones((dout, dout))
    nu, iK, bc = 0.0, None, 0
    if tname.startswith('tpq'):
        nu, iK = float(g[key + '_nu']), g[key + '_iK']
        bc = 1 if (tname == 'tpq1' and dout > 1) else 0
    return co.make_transform(0, din, dout, g[key + '_pts'], g[key + '_wm'], g[key + '_Wc'], g[key + '_Wcc'], emv, bc,
                             nu, iK, integ)


@pytest.mark.parametrize('name', sorted(MODELS))
def test_c_apply_golden(golden, name):
    g = golden('g3_apply')
    fid, p, sidx, din, dout = MODELS[name]
    means, covs, times = g[name + '_mean'], g[name + '_cov'], g[name + '_time']
    for tname in SIGMA_TF + BQ_TF:
        key = '{}_{}'.format(name, tname)
        if key + '_mf' not in g:
            continue
        t, keep = _c_transform(g, name, tname, din, dout, fid, p, sidx)
        mf, cf, cfx, st = co.apply_batch(t, means, covs, times.astype(float), threads=2)
        assert not st.any()
        for i in range(means.shape[0]):
            assert_moments_close((mf[i], cf[i], cfx[i]), (g[key + '_mf'][i], g[key + '_cf'][i], g[key + '_cfx'][i]),
                                 covs[i], what=(key, i))


def test_c_filter_golden(golden):
    g = golden('g4_filters')
    y = g['ungm_y']                                      # (1, T, seeds)
    par = np.array([1.0, 3.0])
    pts = orc.points_ut(1)
    w = orc.gp_weights(par, pts)
    one = np.ones((1, 1))
    td, k1 = co.make_transform(0, 1, 1, pts, w['wm'], w['Wc'], w['Wcc'], w['model_var'] * one,
                               integrand=co.Integrand.make(orc.F_UNGM_DYN))
    to, k2 = co.make_transform(0, 1, 1, pts, w['wm'], w['Wc'], w['Wcc'], w['model_var'] * one,
                               integrand=co.Integrand.make(orc.F_UNGM_MEAS))
    fm, fP, st = co.filter_forward(td, to, y.transpose(2, 1, 0), np.zeros(1), one, 10.0 * one, one, threads=2)
    assert not st.any()
    assert rel_err(fm.transpose(2, 1, 0), g['ungm_gpqkf_fm']) < 1e-9
    assert cov_err(fP.transpose(2, 3, 1, 0), g['ungm_gpqkf_fc']) < 1e-8
    # reentry UKF
    y = g['rer_y']
    pts = orc.points_ut(5)
    wm, wc = orc.weights_ut(5)
    td, k3 = co.make_transform(1, 5, 5, pts, wm, wc, integrand=co.Integrand.make(orc.F_REENTRY2D_DYN, (0.1,)))
    to, k4 = co.make_transform(1, 5, 2, pts, wm, wc, integrand=co.Integrand.make(orc.F_RADAR2D_MEAS, (0.0, 0.0)))
    G = g['rer_G']
    fm, fP, st = co.filter_forward(td, to, y.transpose(2, 1, 0), g['rer_m0'], g['rer_P0'],
                                   G.dot(g['rer_Q']).dot(G.T), g['rer_R'])
    assert not st.any()
    # every state row against its own magnitude, every covariance entry against sqrt(P_ii P_jj): the 1e-6-sized
    # position / velocity blocks count as much as the O(1) variance of the ballistic parameter (measured 6e-10 / 1.5e-9)
    assert mean_err(fm.transpose(2, 1, 0), g['rer_ukf_fm']) < 1e-8
    assert cov_err(fP.transpose(2, 3, 1, 0), g['rer_ukf_fc']) < 1e-8


def test_c_not_pd():
    pts = orc.points_ut(2)
    wm, wc = orc.weights_ut(2)
    t, keep = co.make_transform(1, 2, 2, pts, wm, wc, integrand=co.Integrand.make(orc.F_PENDULUM_DYN, (0.01,)))
    covs = np.stack([np.eye(2), np.array([[1.0, 2.0], [2.0, 1.0]])])
    mf, cf, cfx, st = co.apply_batch(t, np.zeros((2, 2)), covs, 0.0)
    assert list(st) == [0, 1]
