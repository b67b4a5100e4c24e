#!/usr/bin/env python3
"""Smoother on the reentry (5, 2, 11) shape, B = 1e5, T = 50, device-resident (run under rocprofv3 --kernel-trace)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from ssmtoybox_amd import _lib, ssinf, ssmod as sm  # noqa: E402
from ssmtoybox_amd.mtran import resolve_integrand  # noqa: E402
from bench import simulate_reentry  # noqa: E402

amd.set_device(0)
lib = _lib.load()
B, T, D, Y = 100000, 50, 5, 2
x, y, m0, P0, Q, G, R = simulate_reentry(B, T, 3, False)
ld = (B + 63) // 64 * 64
dyn = sm.ReentryVehicle2DTransition(sm.GaussRV(5, m0, P0), sm.GaussRV(3, cov=Q))
obs = sm.Radar2DMeasurement(sm.GaussRV(2, cov=R), 5)
alg = ssinf.UnscentedKalman(dyn, obs)
yb = np.zeros((T, Y, ld)); yb[:, :, :B] = y.transpose(1, 0, 2)
d_y = _lib.DeviceBuffer(yb.nbytes); d_y.upload(yb)
mb = np.zeros((D, ld)); mb[:] = m0[:, None]
Pb = np.zeros((D * D, ld)); Pb[:] = P0.reshape(-1, 1)
d_m0, d_P0 = _lib.DeviceBuffer(mb.nbytes), _lib.DeviceBuffer(Pb.nbytes)
d_m0.upload(mb); d_P0.upload(Pb)
bufs = [_lib.DeviceBuffer(8 * T * n * ld) for n in (D, D * D, D, D * D)]
d_st = _lib.DeviceBuffer(4 * ld)
f_dyn, e_dyn = resolve_integrand(dyn.dyn_eval)
f_obs, e_obs = resolve_integrand(obs.meas_eval)
h_dyn, h_obs = alg.tf_dyn._handle_for(e_dyn), alg.tf_obs._handle_for(e_obs)
gqg, pg = _lib.as_c(G.dot(Q).dot(G.T)); rr, pr = _lib.as_c(R)
for _ in range(4):
    _lib.check(lib.ssmq_filter_smooth_dev(ctypes.c_void_p(h_dyn), ctypes.byref(f_dyn), ctypes.c_void_p(h_obs),
                                          ctypes.byref(f_obs), B, ld, T, ctypes.c_void_p(d_y.ptr), ctypes.c_void_p(d_m0.ptr),
                                          ctypes.c_void_p(d_P0.ptr), pg, pr, *[ctypes.c_void_p(b.ptr) for b in bufs],
                                          ctypes.c_void_p(d_st.ptr)), 'ssmq_filter_smooth_dev')
print('done; failed trajectories', int((d_st.download((ld,), dtype=np.int32)[:B] != 0).sum()))
