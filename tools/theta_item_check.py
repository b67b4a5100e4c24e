#!/usr/bin/env python3
"""k_theta_item (one lane per item) against the two-launch route (k_theta_weights + k_theta_chain) on the same items."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from ssmtoybox_amd import ssinf, ssmod as sm  # noqa: E402

amd.set_device(0)
rng = np.random.default_rng(3)
systems = {
    'ungm sr': (sm.UNGMTransition(sm.GaussRV(1), sm.GaussRV(1, cov=np.array([[10.0]]))), sm.UNGMMeasurement(sm.GaussRV(1), 1), 'sr'),
    'ungm ut': (sm.UNGMTransition(sm.GaussRV(1), sm.GaussRV(1, cov=np.array([[10.0]]))), sm.UNGMMeasurement(sm.GaussRV(1), 1), 'ut'),
    'ungmna sr': (sm.UNGMNATransition(sm.GaussRV(1), sm.GaussRV(1, cov=np.array([[10.0]]))), sm.UNGMMeasurement(sm.GaussRV(1), 1), 'sr'),
    'pendulum sr': (sm.Pendulum2DTransition(sm.GaussRV(2, mean=np.array([1.5, 0.0]), cov=0.01 * np.eye(2)), sm.GaussRV(2, cov=0.01 * np.eye(2)), 0.01),
                    sm.Pendulum2DMeasurement(sm.GaussRV(1, cov=np.array([[0.1]])), 2), 'sr'),
    'pendulum ut': (sm.Pendulum2DTransition(sm.GaussRV(2, mean=np.array([1.5, 0.0]), cov=0.01 * np.eye(2)), sm.GaussRV(2, cov=0.01 * np.eye(2)), 0.01),
                    sm.Pendulum2DMeasurement(sm.GaussRV(1, cov=np.array([[0.1]])), 2), 'ut'),
}
for name, (dyn, obs, pts) in systems.items():
    alg = ssinf.MarginalizedGaussianProcessKalman(dyn, obs, 'rbf', pts)
    D = dyn.dim_state
    n = 2000
    theta = 0.8 * rng.standard_normal((n, alg.param_dim))
    theta[::7] *= 4.0                                       # some far out
    m = rng.standard_normal((n, D)) * 2.0
    a = rng.standard_normal((n, D, D))
    P = np.einsum('nij,nkj->nik', a, a) + 0.3 * np.eye(D)
    y = rng.standard_normal((n, obs.dim_out))
    res = {}
    for mode in ('item', 'two'):
        if mode == 'two':
            os.environ['SSMQ_NO_THETA_ITEM'] = '1'
        else:
            os.environ.pop('SSMQ_NO_THETA_ITEM', None)
        res[mode] = alg.theta_step(theta, m, P, y, 3)
    os.environ.pop('SSMQ_NO_THETA_ITEM', None)
    (m1, c1, l1, s1), (m2, c2, l2, s2) = res['item'], res['two']
    ok = (s1 == 0) & (s2 == 0)

    def rel(a_, b_):
        d = np.abs(a_ - b_) / np.maximum(np.abs(b_), 1e-300)
        return float(np.nanmax(d)) if d.size else 0.0
    print('%-12s items %d  flags equal %s (%d flagged)  bitwise equal: m %s c %s ll %s   max rel diff m %.2e c %.2e ll %.2e' % (
        name, n, np.array_equal(s1, s2), int((s2 != 0).sum()), np.array_equal(m1[ok], m2[ok]), np.array_equal(c1[ok], c2[ok]),
        np.array_equal(l1[ok], l2[ok]), rel(m1[ok], m2[ok]), rel(c1[ok], c2[ok]), rel(l1[ok], l2[ok])))
    if not np.array_equal(s1, s2):
        bad = np.flatnonzero(s1 != s2)[:5]
        print('   flag differences at', bad, s1[bad], s2[bad], 'theta', theta[bad])
