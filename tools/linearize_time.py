#!/usr/bin/env python3
"""k_linearize (LinearizationTransform, csrc/ssmq_linear.hip) on device-resident planes: time per launch and the HBM rate on
the algorithmic bytes 8 (D + D^2 + E + E^2 + E D) per trajectory, for the pendulum dynamics (D = E = 2) and the constant-
velocity model (D = E = 4) at B = 1e6."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from ssmtoybox_amd import _lib, ssmod as sm  # noqa: E402

amd.set_device(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
for name, mod in (('pendulum 2-D', sm.Pendulum2DTransition(sm.GaussRV(2), sm.GaussRV(2), dt=0.01)),
                  ('constant velocity 4-D', sm.ConstantVelocity(sm.GaussRV(4), sm.GaussRV(2), dt=0.5)),
                  ('UNGM 1-D', sm.UNGMTransition(sm.GaussRV(1), sm.GaussRV(1)))):
    D = mod.dim_in
    E = mod.dim_state
    tf = amd.LinearizationTransform(D)
    rng = np.random.default_rng(1)
    means = rng.standard_normal((B, D))
    a = rng.standard_normal((B, D, D))
    covs = np.einsum('bij,bkj->bik', a, a) + 0.2 * np.eye(D)
    mean, cov = _lib.SoA.from_host(means), _lib.SoA.from_host(covs)
    mf, cf, cfx = _lib.SoA(E, B), _lib.SoA(E * E, B), _lib.SoA(E * D, B)
    st = _lib.DeviceBuffer(4 * mean.ld)
    tbuf = _lib.DeviceBuffer(8)
    tbuf.upload(np.zeros(1))
    f = mod.dyn_eval
    for _ in range(5):
        tf.apply_batch_dev(f, mean, cov, tbuf, mf, cf, cfx, st, 0)
    _lib.sync()
    e0, e1 = _lib.Event(), _lib.Event()
    e0.record()
    for _ in range(20):
        tf.apply_batch_dev(f, mean, cov, tbuf, mf, cf, cfx, st, 0)
    e1.record()
    _lib.sync()
    ms = e0.elapsed_ms(e1) / 20
    nbytes = 8.0 * B * (D + D * D + E + E * E + E * D)
    print('%s: B = %d, %.1f us per launch, %.0f GB/s on %d algorithmic bytes per trajectory = %.2f of 8 TB/s' % (
        name, B, 1e3 * ms, nbytes / (ms * 1e-3) / 1e9, int(nbytes / B), nbytes / (ms * 1e-3) / 8e12), flush=True)
