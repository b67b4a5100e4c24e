#!/usr/bin/env python3
"""The unisolvent half of BASELINE configs[4] as SURVEY 8d restates it: Bayes-Sard transform, D = E = 10, unscented
points N = 21 = number of basis functions ([0 | I | 2 I] multi-indices), device-resident moments, device integrand."""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from ssmtoybox_amd import _lib, ssmod  # noqa: E402

amd.set_device(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
D = 10
mi = np.hstack((np.zeros((D, 1), dtype=int), np.eye(D, dtype=int), 2 * np.eye(D, dtype=int)))
tf = amd.BayesSardTransform(D, D, np.array([[1.0] + [3.0] * D]), multi_ind=mi, point_str='ut')
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
for _ in range(3):
    tf.apply_batch_dev(f, mean, cov, tbuf, mf, cf, cfx, st, 0)
_lib.sync()
ts = []
for rep in range(5):
    e0, e1 = _lib.Event(), _lib.Event()
    e0.record()
    for _ in range(20):
        tf.apply_batch_dev(f, mean, cov, tbuf, mf, cf, cfx, st, 0)
    e1.record()
    _lib.sync()
    ts.append(e0.elapsed_ms(e1) / 20)
alg = 8 * (D + D * D + D + D * D + D * D) * B
print('B=%d N=21: %.3f ms per launch, %.3e transforms/s, %.0f GB/s algorithmic' % (B, min(ts), B / (min(ts) * 1e-3),
                                                                              alg / (min(ts) * 1e-3) / 1e9))
