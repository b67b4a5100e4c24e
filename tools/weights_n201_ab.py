#!/usr/bin/env python3
"""Quadrature weights at N = 201 (degree-5 rule at D = 10): the LDS-staged route of rounds 2-3 against the many-workgroup route
of round 4 (SSMQ_WEIGHTS_NO_LDS=1 sends this size there) - wall clock of the transform constructors.  Round 4: 3.9 / 2.1 ms
(Bayes-Sard / GP) staged, 3.6 / 1.8 ms many-workgroup."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd
from ssmtoybox_amd.bq.bqmod import n_sum_k
amd.set_device(0)
D = 10
mi = np.hstack([n_sum_k(D, k) for k in range(3)])
par = np.array([[1.0] + [3.0] * D])
for env in (None, '1'):
    if env: os.environ['SSMQ_WEIGHTS_NO_LDS'] = env
    else: os.environ.pop('SSMQ_WEIGHTS_NO_LDS', None)
    for kind in ('bs', 'gp'):
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            tf = amd.BayesSardTransform(D, D, par, mi, 'fs', {'degree': 5}) if kind == 'bs' else amd.GaussianProcessTransform(D, D, par, 'rbf', 'fs', {'degree': 5})
            ts.append(time.perf_counter() - t0)
        print('N=201', kind, 'many-workgroup' if env else 'staged (LDS)', '%.2f ms' % (1e3 * min(ts)), flush=True)
