"""Exploration for tests/test_gpu_parity.py::test_config3_gpqkf_*: how do the fused GPQ-Kalman loop and the C oracle
differ on the 6-D reentry-shaped model (failing steps, pre-failure moments)?  Run on the GPU box."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle as co, ssmq_oracle as orc          # noqa: E402
from bench import simulate_reentry                              # noqa: E402
from ssmtoybox_amd import ssinf, ssmod as sm                    # noqa: E402
from tests.test_gpu_parity import _c_bq_transform               # noqa: E402

B, T = int(os.environ.get('B', 20000)), 50
x, y, m0, P0, Q, G, R = simulate_reentry(B, T, 12, True)
dyn = sm.ReentryVehicle2DBiasTransition(sm.GaussRV(6, m0, P0), sm.GaussRV(4, cov=Q))
obs = sm.Radar2DMeasurement(sm.GaussRV(2, cov=R), 6)
for ell in (3.0, 25.0):
    par = np.array([[1.0] + [ell] * 6])
    runs = {}
    for fast in (True, False):
        if fast:
            os.environ.pop('SSMQ_NO_FASTPATH', None)
        else:
            os.environ['SSMQ_NO_FASTPATH'] = '1'
        gpq = ssinf.GaussianProcessKalman(dyn, obs, par, par)
        fm, fP = gpq.forward_pass_batch(y, raise_on_failure=False)
        runs['fast' if fast else 'dense'] = (fm, fP, gpq.status.copy())
        print(ell, gpq.kernel_name())
    td, k1 = _c_bq_transform(gpq.tf_dyn, 6, co.Integrand.make(orc.F_REENTRY2D_BIAS_DYN, (0.1,)))
    to, k2 = _c_bq_transform(gpq.tf_obs, 2, co.Integrand.make(orc.F_RADAR2D_MEAS, (0.0, 0.0)))
    cfm, cfP, cst = co.filter_forward(td, to, np.ascontiguousarray(y.transpose(2, 1, 0)), m0, P0, G.dot(Q).dot(G.T), R,
                                      threads=16)
    runs['oracle'] = (cfm.transpose(2, 1, 0), cfP.transpose(2, 3, 1, 0), cst)
    for k, (fm, fP, st) in runs.items():
        h = np.bincount(np.where(st > 0, st, T + 1), minlength=T + 2)
        print(' ', k, 'failing-step histogram (1..8, never):', h[1:9], h[T + 1], 'median', np.median(st))
    for a, b in (('fast', 'oracle'), ('dense', 'oracle'), ('fast', 'dense')):
        fa, Pa, sa = runs[a]
        fb, Pb, sb = runs[b]
        na, nb = np.where(sa > 0, sa - 1, T), np.where(sb > 0, sb - 1, T)
        nok = np.minimum(na, nb)
        print('  %s vs %s: failing step equal %.4f, |diff|<=1: %.4f' % (a, b, np.mean(sa == sb), np.mean(np.abs(na - nb) <= 1)))
        for step in range(4):
            msk = nok > step
            if not msk.any():
                continue
            d = np.sqrt(np.abs(Pb[np.arange(6), np.arange(6), step][:, msk]))
            eP = np.abs(Pa[:, :, step][:, :, msk] - Pb[:, :, step][:, :, msk]) / (d[:, None] * d[None, :])
            em = np.abs(fa[:, step][:, msk] - fb[:, step][:, msk]) / d          # in standard deviations
            print('    step %d (n=%d): cov entry-scaled median %.2e q99 %.2e max %.2e | mean err / sigma median %.2e max %.2e' %
                  (step, msk.sum(), np.median(eP.max(axis=(0, 1))), np.quantile(eP.max(axis=(0, 1)), 0.99), eP.max(),
                   np.median(em.max(axis=0)), em.max()))
    for a, b in (('fast', 'oracle'),):
        fa, Pa, sa = runs[a]; fb, Pb, sb = runs[b]
        nok = np.minimum(np.where(sa > 0, sa - 1, T), np.where(sb > 0, sb - 1, T))
        from tests._cases import mean_err, cov_err
        mask = np.arange(T)[:, None] < nok[None, :]
        print('    helper: mean_err %.3e cov_err %.3e' % (mean_err(fa, fb, mask), cov_err(Pa, Pb, mask)))
        dm = np.abs(fa - fb)[:, mask]
        i = np.unravel_index(np.nanargmax(dm), dm.shape)
        print('    worst abs mean diff', dm[i], 'row', i[0], 'ref value', fb[:, mask][i], 'row maxima', np.max(np.abs(fb[:, mask]), axis=1))
