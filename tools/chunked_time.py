#!/usr/bin/env python3
"""k_filter_chunked (the block-steps of a batch in equal strips, one wave per strip, csrc/ssmq_filter_chunked.hip) against
k_filter_fused (a wave keeps its trajectories for all T steps) on the same device-resident batches: time per pass (HIP events), and
whether the results are the same BITS (they must be: a piece starts from the state its predecessor handed over)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from benchlib.common import settle, timed_passes  # noqa: E402
from benchlib.workloads import FilterBench  # noqa: E402

amd.set_device(0)
cases = [('reentry5', 'ukf', 100000, 50), ('reentry5', 'bsqkf', 100000, 50), ('reentry6', 'ukf', 100000, 50), ('reentry5', 'gpqkf', 100000, 50),
         ('reentry5', 'ukf', 70000, 50), ('reentry5', 'ukf', 140000, 50), ('reentry5', 'ukf', 200000, 50), ('reentry5', 'ukf', 12500, 50),
         ('ct', 'ukf', 100000, 20)]
modes = [('0', 'whole pass'), ('1', 'strips, auto'), ('512', '512 strips'), ('768', '768 strips'), ('1024', '1024 strips'), ('1400', '1400 strips')]
if len(sys.argv) > 1:
    cases = cases[:int(sys.argv[1])]
for wl_name, filt, B, T in cases:
    res = {}
    for mode, label in modes:
        if mode:
            os.environ['SSMQ_FUSED_CHUNKED'] = mode
        else:
            os.environ.pop('SSMQ_FUSED_CHUNKED', None)
        wl = FilterBench(amd, B, T, 31, wl_name, filt)
        name = wl.alg.kernel_name(B)
        settle(wl.step, wl._lib.sync)
        ms = min(timed_passes(wl, 3, 20) for _ in range(3))
        res[mode] = (ms,) + tuple(wl.results()) + (name,)
        wl.free()
    os.environ.pop('SSMQ_FUSED_CHUNKED', None)
    ms0, fm0, fP0, st0, n0 = res['0']
    print('%s %s B=%d T=%d   failed %d' % (wl_name, filt, B, T, int((st0 != 0).sum())), flush=True)
    for mode, label in modes:
        ms, fm, fP, st, name = res[mode]
        same = np.array_equal(fm, fm0, equal_nan=True) and np.array_equal(fP, fP0, equal_nan=True) and np.array_equal(st, st0)
        print('   %-13s %.4f ms  (x %.2f)  %s  %s' % (label, ms, ms0 / ms, 'same bits' if same else 'DIFFERENT RESULTS', name[:24]), flush=True)
