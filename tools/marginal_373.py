#!/usr/bin/env python3
"""Trajectory 373 of the bench's UNGM batch step by step through the serial marginalised filter: where and why it fails
(the reference completes this sequence: tests/golden/g14_marginal_failures.npz)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from ssmtoybox_amd import ssinf, ssmod as sm  # noqa: E402
from bench import simulate_ungm  # noqa: E402

np.set_printoptions(precision=6, linewidth=160)
amd.set_device(0)
dyn = sm.UNGMTransition(sm.GaussRV(1), sm.GaussRV(1, cov=np.array([[10.0]])))
obs = sm.UNGMMeasurement(sm.GaussRV(1), 1)
alg = ssinf.MarginalizedGaussianProcessKalman(dyn, obs, 'rbf', 'sr')
b = int(sys.argv[1]) if len(sys.argv) > 1 else 373
_, y = simulate_ungm(1024, 10, 5)
alg.reset()
for k in range(1, 11):
    yk = y[k - 1, b:b + 1]
    m0, P0, pm0, pc0 = alg.x_mean_fi.copy(), alg.x_cov_fi.copy(), alg.param_mean.copy(), alg.param_cov.copy()
    try:
        alg._measurement_update(yk, k)
        print('step', k, 'ok: x', alg.x_mean_fi, alg.x_cov_fi.ravel(), 'theta', alg.param_mean, 'diag cov', np.diag(alg.param_cov))
    except np.linalg.LinAlgError as e:
        print('step', k, 'FAILS:', e)
        print(' state in: m', m0, 'P', P0.ravel(), 'y', yk)
        print(' prior: mean', pm0, '\n cov\n', pc0)
        print(' Laplace: mean', alg.param_mean, '\n cov\n', alg.param_cov, '\n eig', np.linalg.eigvalsh(0.5 * (alg.param_cov + alg.param_cov.T)))
        chol = np.linalg.cholesky(alg.param_cov)
        pts = alg.param_mean[:, None] + chol.dot(alg.param_upts)
        alg.x_mean_fi, alg.x_cov_fi = m0, P0
        mm, cc, ll, st = alg.theta_step(pts.T, m0, P0, yk, k)
        for j in range(pts.shape[1]):
            print('  point', j, pts[:, j], 'exp', np.exp(pts[:, j]), 'status', st[j], 'm', mm[j], 'c', cc[j].ravel(), 'll', ll[j])
        np.savez('gpurun_out/marginal_373.npz', m0=m0, P0=P0, y=yk, k=k, pm0=pm0, pc0=pc0, lap_mean=alg.param_mean, lap_cov=alg.param_cov, pts=pts, st=st)
        break
