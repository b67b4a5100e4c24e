"""Wall clock of forward_pass_batch (host arrays in and out) over repeated calls: looks for the multi-millisecond stalls
that transfers from / to freshly allocated pageable arrays showed in ssmq_apply_batch before it got pinned staging."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssmtoybox_amd import ssinf, ssmod as sm   # noqa: E402
from bench import simulate_ungm                # noqa: E402

dyn = sm.UNGMTransition(sm.GaussRV(1), sm.GaussRV(1, cov=np.array([[10.0]])))
obs = sm.UNGMMeasurement(sm.GaussRV(1), 1)
par = np.array([[1.0, 3.0]])
alg = ssinf.GaussianProcessKalman(dyn, obs, par, par)
for B, T in ((1, 100), (64, 100), (1000, 100), (10000, 100), (12000, 100), (16384, 100), (20000, 100), (100000, 100)):
    x, y = simulate_ungm(B, T, 1)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter()
        alg.forward_pass_batch(y[None])
        ts.append((time.perf_counter() - t0) * 1e3)
    print('UNGM gpqkf B=%6d T=%d ms per call: %s' % (B, T, ' '.join('%.2f' % t for t in ts)), flush=True)
