#!/usr/bin/env python3
"""Why the batched marginalised filter parked a trajectory: replays the mixture of the failing step from the batch's own outputs."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from ssmtoybox_amd import ssinf, ssmod as sm  # noqa: E402
from bench import simulate_ungm  # noqa: E402

np.set_printoptions(precision=5, linewidth=170)
amd.set_device(0)
dyn = sm.UNGMTransition(sm.GaussRV(1), sm.GaussRV(1, cov=np.array([[10.0]])))
obs = sm.UNGMMeasurement(sm.GaussRV(1), 1)
alg = ssinf.MarginalizedGaussianProcessKalman(dyn, obs, 'rbf', 'sr')
_, y = simulate_ungm(1024, 10, 5)
fm, fP = alg.forward_pass_batch(np.ascontiguousarray(y[None]))
for b in np.flatnonzero(alg.batch_failed):
    k = int(alg.batch_failed[b])
    m = fm[:, k - 2, b] if k > 1 else alg.x0_mean
    P = fP[:, :, k - 2, b] if k > 1 else alg.x0_cov
    mean, cov = alg.batch_param_mean[b], alg.batch_param_cov[b]
    print('trajectory', b, 'step', k, 'reason', alg.batch_failed_reason[b], 'state', m, P.ravel(), 'y', y[k - 1, b])
    print(' Laplace mean', mean, 'eig', np.linalg.eigvalsh(0.5 * (cov + cov.T)))
    pts = mean[:, None] + np.linalg.cholesky(cov).dot(alg.param_upts)
    mm, cc, ll, st = alg.theta_step(pts.T, m, P, y[k - 1, b:b + 1], k)
    for j in range(pts.shape[1]):
        print('  point', j, pts[:, j], 'exp', np.exp(np.clip(pts[:, j], -700, 700)), 'status', st[j], 'm', mm[j], 'c', cc[j].ravel())
