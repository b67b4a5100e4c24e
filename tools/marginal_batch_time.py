#!/usr/bin/env python3
"""Batched marginalised filter (forward_pass_batch: lock-step BFGS, csrc/ssmq_marginal.hip) against the per-trajectory loop on
UNGM: seconds per time step at B = 1 (serial), B = 64 / 1024 (batched)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from ssmtoybox_amd import ssinf, ssmod as sm  # noqa: E402
from bench import simulate_ungm  # noqa: E402

amd.set_device(0)
dyn = sm.UNGMTransition(sm.GaussRV(1), sm.GaussRV(1, cov=np.array([[10.0]])))
obs = sm.UNGMMeasurement(sm.GaussRV(1), 1)
alg = ssinf.MarginalizedGaussianProcessKalman(dyn, obs, 'rbf', 'sr')
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10
_, y = simulate_ungm(1024, T, 5)
data = np.ascontiguousarray(y[None])
alg.forward_pass_serial(data[:, :, :1])
t0 = time.perf_counter()
alg.forward_pass_serial(data[:, :, :4])
t_serial = (time.perf_counter() - t0) / 4 / T
print('serial: %.2f ms per trajectory and step' % (1e3 * t_serial))
for B in (1, 64, 1024):
    alg.forward_pass_batch(data[:, :, :B])
    t0 = time.perf_counter()
    alg.forward_pass_batch(data[:, :, :B])
    dt = (time.perf_counter() - t0) / T
    print('batched B=%5d: %.2f ms per step = %.1f x the serial cost of ONE trajectory; %.1f us per trajectory-step; %s' % (
        B, 1e3 * dt, dt / t_serial, 1e6 * dt / B, alg.batch_stats))
