#!/usr/bin/env python3
"""GP quadrature weights at the upper end of the supported point counts (N = 2500 and SSMQ_MAX_PTS = 4096 random points at D = 4)
through the many-workgroup route (csrc/ssmq_weights.hip: k_wb_*, 64-slot inverse) against the oracle: wall clock and errors next
to the 64 cond eps bar."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd
from ssmtoybox_amd.bq.bqkern import device_gp_weights
from oracle import ssmq_oracle as orc
amd.set_device(0)
rng = np.random.default_rng(3)
for N in (2500, 4096):
    pts = rng.standard_normal((4, N))
    par = np.array([[1.0, 0.35, 0.4, 0.3, 0.45]])
    t0 = time.time(); w = device_gp_weights(pts, par); t1 = time.time()
    ref = orc.gp_weights(par[0], pts)
    K = orc.rbf_eval(par[0], pts, scaling=False) + 1e-8 * np.eye(N)
    cond = np.linalg.cond(K)
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    print(N, 'device %.2f s' % (t1 - t0), 'cond %.2e' % cond, 'wm', rel(w['wm'][0], ref['wm']), 'Wc', rel(w['Wc'][0], ref['Wc']), 'Wcc', rel(w['Wcc'][0], ref['Wcc']), 'bar', 64 * cond * 2.2e-16, flush=True)
