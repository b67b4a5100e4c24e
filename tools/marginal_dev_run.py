#!/usr/bin/env python3
"""One call of the batched marginalised filter on the bench's UNGM batch (for rocprofv3 --kernel-trace --stats)."""
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
T, B = 10, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
_, y = simulate_ungm(B, T, 5)
data = np.ascontiguousarray(y[None])
alg.forward_pass_batch(data[:, :, :64])
for _ in range(2):
    t0 = time.perf_counter()
    alg.forward_pass_batch(data)
    dt = time.perf_counter() - t0
    print('%.2f ms per call, %.2f us per trajectory-step, %s, failed %s reasons %s' % (
        1e3 * dt, 1e6 * dt / (B * T), alg.batch_stats, np.flatnonzero(alg.batch_failed).tolist(),
        alg.batch_failed_reason[alg.batch_failed > 0].tolist()))
