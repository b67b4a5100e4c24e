#!/usr/bin/env python3
"""RTS smoother (forward pass that keeps the predictive moments + backward pass) on device-resident measurements:
one-kernel forward pass vs the launch loop (SSMQ_NO_FUSED=1).  Wall clock around the synchronous C call."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from ssmtoybox_amd import _lib, ssinf, ssmod as sm  # noqa: E402
from ssmtoybox_amd.mtran import resolve_integrand  # noqa: E402

amd.set_device(0)
lib = _lib.load()
B, T = 10000, 100
dyn = sm.UNGMTransition(sm.GaussRV(1), sm.GaussRV(1, cov=np.array([[10.0]])))
obs = sm.UNGMMeasurement(sm.GaussRV(1), 1)
par = np.array([[1.0, 3.0]])
alg = ssinf.GaussianProcessKalman(dyn, obs, par, par, 'rbf', 'ut')
d_x, d_y, ld = sm.simulate_dev(dyn, obs, T, B, seed=2)
m0 = np.zeros((1, ld)); P0 = np.ones((1, ld))
d_m0, d_P0 = _lib.DeviceBuffer(m0.nbytes), _lib.DeviceBuffer(P0.nbytes)
d_m0.upload(m0); d_P0.upload(P0)
bufs = [_lib.DeviceBuffer(8 * T * ld) for _ in range(4)]
d_st = _lib.DeviceBuffer(4 * ld)
f_dyn, e_dyn = resolve_integrand(dyn.dyn_eval)
f_obs, e_obs = resolve_integrand(obs.meas_eval)
h_dyn, h_obs = alg.tf_dyn._handle_for(e_dyn), alg.tf_obs._handle_for(e_obs)
gqg, pg = _lib.as_c(10.0 * np.eye(1)); rr, pr = _lib.as_c(np.eye(1))


def run():
    _lib.check(lib.ssmq_filter_smooth_dev(ctypes.c_void_p(h_dyn), ctypes.byref(f_dyn), ctypes.c_void_p(h_obs),
                                          ctypes.byref(f_obs), B, ld, T, ctypes.c_void_p(d_y.ptr), ctypes.c_void_p(d_m0.ptr),
                                          ctypes.c_void_p(d_P0.ptr), pg, pr, *[ctypes.c_void_p(b.ptr) for b in bufs],
                                          ctypes.c_void_p(d_st.ptr)), 'ssmq_filter_smooth_dev')


for label, env in (('fused', None), ('loop', '1')):
    if env:
        os.environ['SSMQ_NO_FUSED'] = env
    else:
        os.environ.pop('SSMQ_NO_FUSED', None)
    run(); run()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); run(); ts.append(time.perf_counter() - t0)
    print('UNGM GPQ smoother B=%d T=%d %-6s: %.3f ms per call (%.2e smoothed steps/s)' % (B, T, label, 1e3 * min(ts), B * T / min(ts)))
