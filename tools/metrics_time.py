#!/usr/bin/env python3
"""Time the Monte-Carlo error-statistics kernels on filter-shaped buffers: run under
`rocprofv3 --kernel-trace --stats` for the per-kernel durations; prints algorithmic bytes per launch."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from ssmtoybox_amd import _lib, mcshard  # noqa: E402

amd.set_device(0)
rng = np.random.default_rng(0)
for D, T, B in ((1, 100, 10000), (5, 50, 100000), (6, 50, 100000)):
    ld = (B + 63) // 64 * 64
    x = rng.standard_normal((T, D, ld))
    m = x + 0.3 * rng.standard_normal((T, D, ld))
    P = np.zeros((T, D * D, ld))
    P[:, ::D + 1, :] = 0.5 + rng.random((T, D, ld))
    bufs = []
    for a in (x, m, P):
        d = _lib.DeviceBuffer(a.nbytes)
        d.upload(a)
        bufs.append(d)
    for _ in range(5):
        s = mcshard.device_error_sums(D, B, ld, T, *bufs)
        l = mcshard.device_lcr_sums(D, B, ld, T, *bufs, s['mse'] / B)
    print('D=%d T=%d B=%d: algorithmic bytes per launch %.1f MB (8 (2 D + D^2) B T)' % (D, T, B, 8 * (2 * D + D * D) * B * T / 1e6))
    for d in bufs:
        d.free()
