"""The oracle refereed in extended precision (no GPU): on the recursions where every fp64 evaluation - the reference's
included - carries per-cent-level rounding noise (uncentred BQ covariance on the reentry model, the t-process filter on
the coordinated-turn model), the oracle must be no further from the EXACT result of the algorithm than the reference's
own NumPy evaluation is.  The GPU suite asks the same of the device (tests/test_gpu_parity.py::test_referee_*)."""
import numpy as np

from oracle import ssmq_oracle as orc
from tests import _referee as rf


def _assert_no_worse(got, ref, what, factor=2.0, floor=1e-12):
    for key, f in (('m_rms', factor), ('P_rms', factor), ('m_max', 1.5 * factor), ('P_max', 1.5 * factor)):
        bad = got[key] > f * ref[key] + floor
        assert not bad.any(), (what, key, np.flatnonzero(bad), got[key][bad], ref[key][bad])


def test_reference_noise_level_on_the_reentry_model():
    """What the referee establishes about the reference itself (bq/bqmtran.py:199): its filtered covariance is off by more
    than 1e-3 (entry-scaled) and its mean by more than a tenth of a posterior standard deviation at the first step."""
    g = rf.load()
    xm, xP = rf.reentry_exact()
    e = rf.step_errors(g['rer_fm'], g['rer_fc'], xm, xP)
    assert e['ok'].all()
    assert 1e-3 < e['P_max'][0] < 0.2 and 0.1 < e['m_max'][0] < 3.0
    assert e['P_max'].max() < 0.2          # ... and it stays at that level: the filter is contractive


def test_oracle_vs_exact_reentry_bsqkf():
    g = rf.load()
    xm, xP = rf.reentry_exact()
    wd, wo = rf.weights(g, 'rer_dyn'), rf.weights(g, 'rer_obs')
    pts, G = g['rer_dyn_pts'], g['rer_G']
    tfd = lambda m, P, t: orc.apply_bq(orc.F_REENTRY2D_DYN, m, P, t, pts, wd, (0.1,))
    tfo = lambda m, P, t: orc.apply_bq(orc.F_RADAR2D_MEAS, m, P, t, pts, wo, (0.0, 0.0))
    y = g['rer_y']
    om, oP = np.zeros_like(xm), np.zeros_like(xP)
    for s in range(y.shape[2]):
        om[..., s], oP[..., s], *_ = orc.gaussian_filter(y[..., s], g['rer_m0'], g['rer_P0'], g['rer_Q'], g['rer_R'], G, tfd, tfo)
    _assert_no_worse(rf.step_errors(om, oP, xm, xP), rf.step_errors(g['rer_fm'], g['rer_fc'], xm, xP), 'oracle, reentry BSQKF')


def test_oracle_vs_exact_ct_tpqkf():
    g = rf.load()
    xm, xP = rf.ct_exact()
    wd, wo = rf.weights(g, 'ct_dyn'), rf.weights(g, 'ct_obs')
    pts, nu, dt = g['ct_dyn_pts'], float(g['ct_nu'][0]), float(g['ct_dt'][0])
    sens = tuple(g['ct_sensors'].reshape(-1))

    def tf(fid, w, p, sidx):
        def apply(m, P, t):
            # StudentProcessKalman: dim_out = 1 transforms, model variance broadcast over the output covariance
            chol = np.linalg.cholesky(P)
            fx = orc.eval_columns(fid, m[:, None] + chol.dot(pts), t, p, sidx)
            S = fx.dot(w['iK']).dot(fx.T)
            emv = (nu - 2 + S) / (nu - 2 + pts.shape[1]) * float(w['model_var'][0, 0])
            mean_f = fx.dot(w['wm'])
            return mean_f, fx.dot(w['Wc']).dot(fx.T) - np.outer(mean_f, mean_f) + emv, fx.dot(w['Wcc'].T).dot(chol.T)
        return apply
    y = g['ct_y']
    om, oP = np.full_like(xm, np.nan), np.full_like(xP, np.nan)
    for s in range(y.shape[2]):
        try:
            om[..., s], oP[..., s], *_ = orc.gaussian_filter(y[..., s], g['ct_m0'], g['ct_P0'], g['ct_Q'], g['ct_R'], np.eye(5),
                                                             tf(orc.F_CT_DYN, wd, (dt,), None),
                                                             tf(orc.F_BEARING_MEAS, wo, sens, (0, 2)))
        except np.linalg.LinAlgError:
            pass
    ref = rf.step_errors(g['ct_fm'], g['ct_fc'], xm, xP)
    got = rf.step_errors(om, oP, xm, xP)
    assert got['ok'].mean() > 0.9 and ref['ok'].mean() > 0.9
    # rounding differences grow from step to step on this model (1000-sized positions through the uncentred covariance)
    assert ref['m_max'][0] < 1e-9 and ref['m_max'][-1] > 3 * ref['m_max'][0]
    _assert_no_worse(got, ref, 'oracle, coordinated-turn TPQKF', factor=3.0)
    print('ct reference', ref['m_max'], ref['P_max'], 'oracle', got['m_max'], got['P_max'])
