#!/usr/bin/env python3
"""Device-resident rounds of the batched marginalised filter against round 4's host rounds (SSMQ_MARGINAL_HOST_ROUNDS=1) and the
serial per-trajectory path on the bench's UNGM batch: time, rounds, agreement, and which trajectories fail where (the failed
ones are written to gpurun_out/marginal_failed.npz with their measurement sequences)."""
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
T, B = 10, 1024
_, y = simulate_ungm(B, T, 5)
data = np.ascontiguousarray(y[None])
res = {}
for mode in ('device', 'host'):
    if mode == 'host':
        os.environ['SSMQ_MARGINAL_HOST_ROUNDS'] = '1'
    else:
        os.environ.pop('SSMQ_MARGINAL_HOST_ROUNDS', None)
    alg.forward_pass_batch(data[:, :, :64])
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        fm, fP = alg.forward_pass_batch(data)
        ts.append(time.perf_counter() - t0)
    dt = min(ts)
    res[mode] = (fm.copy(), fP.copy(), alg.batch_failed.copy(), dict(alg.batch_stats))
    print('%-6s rounds: %.2f ms per call, %.2f us per trajectory-step; stats %s; failed %s at steps %s' % (
        mode, 1e3 * dt, 1e6 * dt / (B * T), alg.batch_stats, np.flatnonzero(alg.batch_failed).tolist(),
        alg.batch_failed[alg.batch_failed > 0].tolist()))
os.environ.pop('SSMQ_MARGINAL_HOST_ROUNDS', None)
fd, Pd, bd, _ = res['device']
fh, Ph, bh, _ = res['host']
both = (bd == 0) & (bh == 0)
sd = np.sqrt(np.abs(Ph[0, 0][:, both]))
dm = np.abs(fd[0][:, both] - fh[0][:, both]) / sd
dP = np.abs(Pd[0, 0][:, both] - Ph[0, 0][:, both]) / np.abs(Ph[0, 0][:, both])
print('device vs host rounds: failed sets equal %s; |dm|/sd first step max %.2e median %.2e p99 %.2e max %.2e; |dP|/P median %.2e p99 %.2e' % (
    np.array_equal(bd, bh), dm[0].max(), np.median(dm), np.quantile(dm, 0.99), dm.max(), np.median(dP), np.quantile(dP, 0.99)))
# the failed trajectories one by one through the serial path (scipy BFGS per step, where forward_pass raises LinAlgError)
idx = np.flatnonzero(bd | bh)
for b in idx:
    try:
        alg.reset()
        alg.forward_pass(data[:, :, b])
        out = 'serial path: no exception'
    except np.linalg.LinAlgError as e:
        out = 'serial path raises LinAlgError: %s' % e
    print('trajectory %d: device rounds fail at step %d, host rounds at %d; %s' % (b, bd[b], bh[b], out))
os.makedirs('gpurun_out', exist_ok=True)
np.savez('gpurun_out/marginal_failed.npz', idx=idx, y=data[:, :, idx], failed_device=bd[idx], failed_host=bh[idx],
         fm_device=fd[:, :, idx], fm_host=fh[:, :, idx])
