#!/usr/bin/env python3
"""Throughput of the filters for models with non-additive noise (UNGMNA, CTRS + radar): one fused kernel against the
launch loop (SSMQ_NO_FUSED=1).  Data generated on the device; wall clock around synchronous calls."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from ssmtoybox_amd import ssinf, ssmod as sm  # noqa: E402

amd.set_device(0)
q = sm.GaussRV(1, cov=np.array([[10.0]]))
cases = {
    'ungmna ukf': (sm.UNGMNATransition(sm.GaussRV(1, mean=np.array([1.0])), q), sm.UNGMNAMeasurement(sm.GaussRV(1), 1), 10000, 100),
    'ctrs ukf': (sm.ConstantTurnRateSpeed(sm.GaussRV(5, mean=np.array([10.0, 10.0, 5.0, 0.3, 0.1]), cov=0.1 * np.eye(5)),
                                          sm.GaussRV(2, cov=np.diag([0.1, 0.1 * np.pi]))),
                 sm.Radar2DMeasurement(sm.GaussRV(2, cov=np.diag([0.3, 0.03])), 5), 100000, 50),
}
for name, (dyn, obs, B, T) in cases.items():
    alg = ssinf.UnscentedKalman(dyn, obs)
    d_x, d_y, ld = sm.simulate_dev(dyn, obs, T, B, seed=3)
    for label, env in (('fused', None), ('loop', '1')):
        if env:
            os.environ['SSMQ_NO_FUSED'] = env
        else:
            os.environ.pop('SSMQ_NO_FUSED', None)
        ts = []
        for r in range(6):
            t0 = time.perf_counter()
            bufs = alg.forward_pass_dev(d_y, B, ld, T)
            ts.append(time.perf_counter() - t0)
            for b in bufs:
                b.free()
        t = min(ts[1:])
        print('%-12s %-6s B=%d T=%d: %.3f ms per pass  %.3e filter steps/s' % (name, label, B, T, 1e3 * t, B * T / t))
    os.environ.pop('SSMQ_NO_FUSED', None)
    d_x.free()
    d_y.free()
