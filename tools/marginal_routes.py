#!/usr/bin/env python3
"""The batched marginalised filter on the bench's UNGM batch by its three routes (one launch | device rounds | host rounds):
time per trajectory-step, work counters, and whether the first two agree bit for bit."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from ssmtoybox_amd import ssinf, ssmod as sm  # noqa: E402
from bench import simulate_ungm  # noqa: E402

amd.set_device(0)
dyn = sm.UNGMTransition(sm.GaussRV(1), sm.GaussRV(1, cov=np.array([[10.0]])))
obs = sm.UNGMMeasurement(sm.GaussRV(1), 1)
alg = ssinf.MarginalizedGaussianProcessKalman(dyn, obs, 'rbf', 'sr')
T = 10
for B in [int(v) for v in sys.argv[1:]] or [1024, 16384]:
    _, y = simulate_ungm(B, T, 5)
    data = np.ascontiguousarray(y[None])
    res = {}
    for route, env in (('one launch', None), ('device rounds', 'SSMQ_MARGINAL_ROUNDS'), ('host rounds', 'SSMQ_MARGINAL_HOST_ROUNDS')):
        if route == 'host rounds' and B > 2048:
            continue
        if env:
            os.environ[env] = '1'
        alg.forward_pass_batch(data[:, :, :64])
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            f, P = alg.forward_pass_batch(data)
            best = min(best, time.perf_counter() - t0)
        if env:
            del os.environ[env]
        res[route] = (f, P, alg.batch_failed.copy())
        print('B %6d  %-14s %9.2f ms  %6.3f us per trajectory-step  %s  failed %d' % (
            B, route, 1e3 * best, 1e6 * best / (B * T), alg.batch_stats, int((alg.batch_failed != 0).sum())), flush=True)
    a, b = res['one launch'], res['device rounds']
    print('B %6d  one launch == device rounds: %s' % (B, all(np.array_equal(u, v, equal_nan=True) for u, v in zip(a, b))), flush=True)
