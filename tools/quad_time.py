#!/usr/bin/env python3
"""k_filter_quad (one trajectory on the four lanes of a quad, sigma points dealt to the lanes) against k_filter_fused (one trajectory
per lane, all points) on the same device-resident batches: time per pass (HIP events) and the difference of the filtered moments."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from benchlib.common import settle, timed_passes  # noqa: E402
from benchlib.workloads import FilterBench  # noqa: E402

amd.set_device(0)
cases = [('reentry5', 'ukf', 12500, 50), ('reentry6', 'ukf', 12500, 50), ('ct', 'ukf', 10000, 20), ('reentry5', 'ukf', 4096, 50), ('reentry5', 'ukf', 16384, 50),
         ('reentry5', 'ukf', 20000, 50), ('reentry5', 'ukf', 32768, 50)]
if len(sys.argv) > 1:
    cases = cases[:int(sys.argv[1])]
for wl_name, filt, B, T in cases:
    res = {}
    for mode in ('0', '1'):
        os.environ['SSMQ_FUSED_QUAD'] = mode
        wl = FilterBench(amd, B, T, 31, wl_name, filt)
        name = wl.alg.kernel_name(B)
        settle(wl.step, wl._lib.sync)
        ms = min(timed_passes(wl, 3, 20) for _ in range(3))
        fm, fP, st = wl.results()
        res[mode] = (ms, fm, fP, st, name)
        wl.free()
    os.environ.pop('SSMQ_FUSED_QUAD')
    ms0, fm0, fP0, st0, n0 = res['0']
    print('%s %s B=%d T=%d' % (wl_name, filt, B, T))
    print('   fused   %.4f ms  failed %d   %s' % (ms0, int((st0 != 0).sum()), n0[:60]))
    for mode in ('1',):
        ms, fm, fP, st, name = res[mode]
        good = (st == 0) & (st0 == 0)
        D = fm.shape[0]
        sd = np.sqrt(np.abs(fP0[np.arange(D), np.arange(D)][:, :, good]))
        dm = np.abs(fm[:, :, good] - fm0[:, :, good]) / sd
        dP = np.abs(fP[:, :, :, good] - fP0[:, :, :, good]) / np.maximum(np.abs(fP0[:, :, :, good]).max(axis=(0, 1), keepdims=True), 1e-300)
        print('   quad=%s   %.4f ms  (x %.2f)  failed %d  status equal %.4f  |dm|/sd first step max %.2e, all steps median %.2e p99 %.2e max %.2e  |dP|/max|P| median %.2e max %.2e  %s' % (
            mode, ms, ms0 / ms, int((st != 0).sum()), float(np.mean((st == 0) == (st0 == 0))), float(dm[:, 0].max()), float(np.median(dm)), float(np.quantile(dm, 0.99)), float(dm.max()),
            float(np.median(dP)), float(dP.max()), name[-24:]))
