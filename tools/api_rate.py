#!/usr/bin/env python3
"""forward_pass_batch at configs[1] (host arrays in and out): the pipelined route against upload / pass / download (SSMQ_NO_PIPED=1),
and the pipelined route's time by block count through the C entry point."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from ssmtoybox_amd import _lib, ssinf, ssmod  # noqa: E402
from ssmtoybox_amd.mtran import resolve_integrand  # noqa: E402
from benchlib.workloads import simulate_ungm  # noqa: E402

amd.set_device(0)
B, T = int(os.environ.get('B', '10000')), int(os.environ.get('T', '100'))
_, y = simulate_ungm(B, T, 1)
y = np.ascontiguousarray(y[None])
dyn = ssmod.UNGMTransition(ssmod.GaussRV(1), ssmod.GaussRV(1, cov=np.array([[10.0]])))
obs = ssmod.UNGMMeasurement(ssmod.GaussRV(1), 1)
par = np.array([[1.0, 3.0]])
alg = ssinf.GaussianProcessKalman(dyn, obs, par, par, 'rbf', 'ut')


def best(fn, reps=15):
    fn()
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return 1e3 * min(ts), 1e3 * float(np.median(ts))


os.environ['SSMQ_NO_PIPED'] = '1'
print('upload / pass / download    min %.3f ms  median %.3f ms' % best(lambda: alg.forward_pass_batch(y, raise_on_failure=False)))
os.environ.pop('SSMQ_NO_PIPED')
print('pipelined (pinned results)  min %.3f ms  median %.3f ms' % best(lambda: alg.forward_pass_batch(y, raise_on_failure=False)))
lib = _lib.load()
f_dyn, e_dyn = resolve_integrand(dyn.dyn_eval)
f_obs, e_obs = resolve_integrand(obs.meas_eval)
dp = lambda a: a.ctypes.data_as(_lib.c_double_p)      # noqa: E731
m0, P0, q, r = np.zeros(1), np.ones((1, 1)), np.array([[10.0]]), np.ones((1, 1))
st = np.zeros(B, dtype=np.int32)
for pinned in (1, 0):
    fm = _lib.pinned_empty((1, T, B)) if pinned else np.empty((1, T, B))
    fP = _lib.pinned_empty((1, 1, T, B)) if pinned else np.empty((1, 1, T, B))
    for K in (1, 2, 3, 4, 5, 7, 10, 16):
        call = lambda: _lib.check(lib.ssmq_filter_forward_piped(      # noqa: E731
            ctypes.c_void_p(alg.tf_dyn._handle_for(e_dyn)), ctypes.byref(f_dyn), ctypes.c_void_p(alg.tf_obs._handle_for(e_obs)), ctypes.byref(f_obs),
            B, T, dp(y), dp(m0), dp(P0), dp(q), dp(r), ctypes.c_void_p(fm.ctypes.data), ctypes.c_void_p(fP.ctypes.data),
            ctypes.c_void_p(st.ctypes.data), pinned, K))
        print('C entry point, %s results, K = %2d blocks   min %.3f ms  median %.3f ms' % ((('pinned  ' if pinned else 'pageable'), K) + best(call)))
