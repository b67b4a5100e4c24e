#!/usr/bin/env python3
"""Host-buffer (PCIe-inclusive) rates of the drop-in entry points, for DESIGN.md section 5: the reference-shaped calls
that take and return NumPy arrays (upload + layout conversion + kernel + download), wall clock."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from ssmtoybox_amd import ssinf, ssmod as sm  # noqa: E402
from bench import simulate_ungm  # noqa: E402

amd.set_device(0)


def best(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


# GPQ-Kalman on UNGM through forward_pass_batch: y (1, T, B) in, (D, T, B) + (D, D, T, B) out
B, T = 10000, 100
x, y = simulate_ungm(B, T, 1)
dyn = sm.UNGMTransition(sm.GaussRV(1), sm.GaussRV(1, cov=np.array([[10.0]])))
obs = sm.UNGMMeasurement(sm.GaussRV(1), 1)
par = np.array([[1.0, 3.0]])
alg = ssinf.GaussianProcessKalman(dyn, obs, par, par, 'rbf', 'ut')
yy = np.ascontiguousarray(y.T[None].transpose(0, 2, 1))
t = best(lambda: alg.forward_pass_batch(yy))
print('forward_pass_batch UNGM GPQKF  B=%d T=%d: %.3f ms per call -> %.3e filter steps/s (host arrays in and out)' % (B, T, 1e3 * t, B * T / t))

# batched GPQ transform D = E = 6 through apply_batch: (B, 6), (B, 6, 6) in, three arrays out
B = 100000
rng = np.random.default_rng(2)
m0 = np.array([6500.4, 349.14, -1.8093, -6.7967, 0.6932, 0.0])
p0 = np.array([1e-6, 1e-6, 1e-6, 1e-6, 1.0, 1e-2])
means = m0 + rng.standard_normal((B, 6)) * np.sqrt(p0)
a = rng.standard_normal((B, 6, 6)) / np.sqrt(6)
s = np.sqrt(p0)
covs = np.einsum('i,bij,bkj,k->bik', s, a, a, s) + 1e-6 * np.diag(p0)
tf = amd.GaussianProcessTransform(6, 6, np.array([[1.0] + [3.0] * 6]), 'rbf', 'ut')
f = sm.ReentryVehicle2DBiasTransition(dt=0.1).dyn_eval
t = best(lambda: tf.apply_batch(f, means, covs, 0.0))
print('apply_batch GPQ D=E=6 N=13     B=%d: %.3f ms per call -> %.3e transforms/s, %.2f GB/s algorithmic (host arrays in and out)' % (
    B, 1e3 * t, B / t, 960.0 * B / t / 1e9))
