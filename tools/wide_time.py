#!/usr/bin/env python3
"""Generic-kernel timing at the BSQ D = 10 shapes (config C5): N = 21 and N = 201, reductions-only entry point
(ssmq_apply_fx_batch) with host-made integrand values.  Run under rocprofv3 --kernel-trace --stats."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from ssmtoybox_amd.bq.bqmod import n_sum_k  # noqa: E402

amd.set_device(0)
rng = np.random.default_rng(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
means = rng.standard_normal((B, 10))
a = rng.standard_normal((B, 10, 10)) / np.sqrt(10)
covs = np.einsum('bij,bkj->bik', a, a) + 0.1 * np.eye(10)


def f(x, par):
    return np.concatenate((np.sin(x[:5]) + x[5:] ** 2, x[5:] * np.cos(x[:5])))


for pstr, ppar, mi in (('ut', None, np.hstack((np.zeros((10, 1)), np.eye(10), 2 * np.eye(10))).astype(int)),
                       ('fs', {'degree': 5}, np.hstack([n_sum_k(10, k) for k in range(3)]))):
    tf = amd.BayesSardTransform(10, 10, np.array([[1.0] + [3.0] * 10]), mi, pstr, ppar)
    t0 = time.perf_counter()
    tf.apply_batch(f, means, covs, 0.0)
    print('N=%d B=%d wall %.2f s (includes the host evaluation of f at B*N points)' % (tf.wm.shape[0], B,
                                                                                      time.perf_counter() - t0))
