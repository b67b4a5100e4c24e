#!/usr/bin/env python3
"""Where the host-inclusive call forward_pass_batch (UNGM GPQ-Kalman, B = 1e4, T = 100) spends its time: allocation, upload,
kernel, download."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssmtoybox_amd import _lib, ssinf, ssmod as sm   # noqa: E402
from bench import simulate_ungm                # noqa: E402

dyn = sm.UNGMTransition(sm.GaussRV(1), sm.GaussRV(1, cov=np.array([[10.0]])))
obs = sm.UNGMMeasurement(sm.GaussRV(1), 1)
par = np.array([[1.0, 3.0]])
alg = ssinf.GaussianProcessKalman(dyn, obs, par, par)
B, T = 10000, 100
x, y = simulate_ungm(B, T, 1)
data = np.ascontiguousarray(y[None])
for _ in range(3):
    alg.forward_pass_batch(data)
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    alg.forward_pass_batch(data)
    ts.append(time.perf_counter() - t0)
print('forward_pass_batch: min %.3f ms median %.3f ms' % (1e3 * min(ts), 1e3 * np.median(ts)))
ld = (B + 63) // 64 * 64


def tm(fn, n=10):
    fn()
    v = []
    for _ in range(n):
        t0 = time.perf_counter()
        r = fn()
        _lib.sync()
        v.append(time.perf_counter() - t0)
    return 1e3 * min(v), r


t_alloc, bufs = tm(lambda: [_lib.DeviceBuffer(8 * T * ld), _lib.DeviceBuffer(8 * ld), _lib.DeviceBuffer(8 * ld), _lib.DeviceBuffer(8 * T * ld),
                            _lib.DeviceBuffer(8 * T * ld), _lib.DeviceBuffer(4 * ld)])
d_y, d_m0, d_P0, d_fm, d_fP, d_st = bufs
t_free, _ = tm(lambda: [b.free() for b in [_lib.DeviceBuffer(8 * T * ld) for _ in range(6)]])
t_up, _ = tm(lambda: _lib.upload_study(data, 1, ld, d_y))
t_dfm, _ = tm(lambda: _lib.download_study(d_fm, (1,), T, B, ld))
t_dfP, _ = tm(lambda: _lib.download_study(d_fP, (1, 1), T, B, ld))
t_st, _ = tm(lambda: d_st.download((ld,), dtype=np.int32))
t_small, _ = tm(lambda: (d_m0.upload(np.zeros((1, ld))), d_P0.upload(np.ones((1, ld)))))
t_empty, _ = tm(lambda: np.empty((1, T, B)))
print('alloc 6 buffers %.3f ms | alloc+free 6 %.3f | upload y %.3f | download fm %.3f | download fP %.3f | status %.3f | m0, P0 %.3f' % (
    t_alloc, t_free, t_up, t_dfm, t_dfP, t_st, t_small))
