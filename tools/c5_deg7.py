#!/usr/bin/env python3
"""BASELINE configs[4] as worded: Bayes-Sard transform, D = E = 10, fully-symmetric rule of degree 7 (this build's rule:
the reference has degree 3 and 5 only, mtran.py:392) = 1181 points, multi-index of total degree <= 2 (66 basis functions),
device integrand: weights on the device, then the evaluation pass + blocked matrix-core GEMM + per-trajectory rest."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from ssmtoybox_amd import _lib, ssmod  # noqa: E402
from ssmtoybox_amd.bq.bqmod import n_sum_k  # noqa: E402
from oracle import ssmq_oracle as orc  # noqa: E402

amd.set_device(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
D = 10
mi = np.hstack([n_sum_k(D, k) for k in range(3)])
t0 = time.time()
tf = amd.BayesSardTransform(D, D, np.array([[1.0] + [3.0] * D]), mi, 'fs', {'degree': 7})
print('weights: N = %d, %.2f s' % (tf.wm.shape[0], time.time() - t0), 'sum wm', tf.wm.sum(), 'model_var', tf.model.model_var)
f = ssmod.Smooth10DTransition().dyn_eval
rng = np.random.default_rng(6)
means = rng.standard_normal((B, D))
a = rng.standard_normal((B, D, D)) / np.sqrt(D)
covs = np.einsum('bij,bkj->bik', a, a) + 0.1 * np.eye(D)
mean, cov = _lib.SoA.from_host(means), _lib.SoA.from_host(covs)
mf, cf, cfx = _lib.SoA(D, B), _lib.SoA(D * D, B), _lib.SoA(D * D, B)
st = _lib.DeviceBuffer(4 * mean.ld)
tbuf = _lib.DeviceBuffer(8)
tbuf.upload(np.zeros(1))
print(tf.kernel_name(f))
tf.apply_batch_dev(f, mean, cov, tbuf, mf, cf, cfx, st, 0)
_lib.sync()
ts = []
for rep in range(3):
    e0, e1 = _lib.Event(), _lib.Event()
    e0.record()
    for _ in range(3):
        tf.apply_batch_dev(f, mean, cov, tbuf, mf, cf, cfx, st, 0)
    e1.record()
    _lib.sync()
    ts.append(e0.elapsed_ms(e1) / 3)
N = tf.wm.shape[0]
flop = 2.0 * B * D * N * N
print('B=%d N=%d: %.3f ms per transform batch, %.3e transforms/s, GEMM-equivalent %.1f TFLOP/s' % (
    B, N, min(ts), B / (min(ts) * 1e-3), flop / (min(ts) * 1e-3) / 1e12))
g_mf, g_cf, g_cfx = mf.to_host(), cf.to_host((D, D)), cfx.to_host((D, D))
w = dict(wm=tf.wm, Wc=tf.Wc, Wcc=tf.Wcc, model_var=tf.model.model_var)
worst = 0.0
for i in (0, B // 2, B - 1):
    r = orc.apply_bq(orc.F_SMOOTH10D_DYN, means[i], covs[i], 0.0, tf.model.points, w)
    s = float(np.max(np.abs(r[0])))
    worst = max(worst, np.max(np.abs(g_mf[i] - r[0])) / s, np.max(np.abs(g_cf[i] - r[1])) / max(s ** 2, np.abs(r[1]).max()),
                np.max(np.abs(g_cfx[i] - r[2])) / max(np.abs(r[2]).max(), s))
print('max scaled error vs oracle on 3 trajectories: %.2e' % worst)
