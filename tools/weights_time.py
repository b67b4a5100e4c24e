#!/usr/bin/env python3
"""Time the theta-batched weights kernel for a few shapes (GP and Bayes-Sard), device time via wall clock around the
synchronous C call (includes H2D/D2H of the small arrays)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from ssmtoybox_amd.bq.bqkern import device_gp_weights  # noqa: E402
from ssmtoybox_amd.bq.bqmod import n_sum_k  # noqa: E402
from ssmtoybox_amd.mtran import UnscentedTransform as UT, FullySymmetricStudentTransform as FS  # noqa: E402

amd.set_device(0)


def t(fn, reps=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


for D, pts, P in ((1, UT.unit_sigma_points(1), 1), (1, UT.unit_sigma_points(1), 4096), (6, UT.unit_sigma_points(6), 1),
                  (6, UT.unit_sigma_points(6), 4096), (10, FS.unit_sigma_points(10, 5), 1),
                  (10, FS.unit_sigma_points(10, 5), 64)):
    par = np.column_stack((np.ones(P), np.linspace(2.0, 4.0, P)[:, None] * np.ones((P, D))))
    dt = t(lambda: device_gp_weights(pts, par))
    print('GP  D=%2d N=%3d P=%5d : %9.3f ms  (%.1f us per theta)' % (D, pts.shape[1], P, 1e3 * dt, 1e6 * dt / P))
for D, pts in ((10, UT.unit_sigma_points(10)), (10, FS.unit_sigma_points(10, 5))):
    mi = np.hstack([n_sum_k(D, k) for k in range(3)]) if pts.shape[1] > 21 else \
        np.hstack((np.zeros((D, 1)), np.eye(D), 2 * np.eye(D))).astype(int)
    par = np.array([[1.0] + [3.0] * D])
    dt = t(lambda: amd.BayesSardTransform(D, D, par, mi, 'ut' if pts.shape[1] == 21 else 'fs',
                                          None if pts.shape[1] == 21 else {'degree': 5}), reps=3)
    print('BS  D=%2d N=%3d NB=%3d      : %9.3f ms' % (D, pts.shape[1], mi.shape[1], 1e3 * dt))
