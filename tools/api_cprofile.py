#!/usr/bin/env python3
"""cProfile of forward_pass_batch (UNGM GPQ-Kalman, B = 1e4, T = 100, host arrays in and out)."""
import cProfile
import os
import pstats
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssmtoybox_amd import ssinf, ssmod as sm   # noqa: E402
from bench import simulate_ungm                # noqa: E402

dyn = sm.UNGMTransition(sm.GaussRV(1), sm.GaussRV(1, cov=np.array([[10.0]])))
obs = sm.UNGMMeasurement(sm.GaussRV(1), 1)
par = np.array([[1.0, 3.0]])
alg = ssinf.GaussianProcessKalman(dyn, obs, par, par)
x, y = simulate_ungm(10000, 100, 1)
data = np.ascontiguousarray(y[None])
for _ in range(3):
    alg.forward_pass_batch(data)
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    alg.forward_pass_batch(data)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
