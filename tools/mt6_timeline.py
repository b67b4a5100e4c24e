#!/usr/bin/env python3
"""Where the launch time of the north-star transform goes: per-wave time stamps of ONE launch of k_apply_small<6,6,13,...> at B = 1e5
(library variant built with -DSSMQ_DIAG_STAMP: tools/build_file_variant.sh ssmq_small_d stamp "-DSSMQ_DIAG_STAMP").
Prints, in microseconds from the first wave's start: when waves start, when their inputs have arrived, when their last store is
issued, when their stores are acknowledged - overall and for SIMDs that hold one / two waves."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from ssmtoybox_amd import _lib  # noqa: E402
from bench import Mt6Bench  # noqa: E402

amd.set_device(0)
B = int(os.environ.get('MT6_B', '100000'))
lpw = int(os.environ.get('MT6_LPW', '64'))
mt = Mt6Bench(amd, B, seed=2)
print('kernel', mt.kernel)
for _ in range(20):
    mt.launch()
_lib.sync()
lib = _lib.load()
nw = (B + lpw - 1) // lpw
for rep in range(3):
    mt.launch()
    _lib.sync()
    buf = np.zeros((nw, 8), dtype=np.uint64)
    rc = lib.ssmq_diag_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(nw))
    assert rc == 0, rc
    t = buf[:, :4].astype(np.int64)
    t0 = t[:, 0].min()
    us = (t - t0) * 0.01                          # 100 MHz counter
    hw = buf[:, 4]
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 15
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    xcc = buf[:, 5] & 15
    key = ((xcc * 8 + se) * 2 + sh) * 64 + cu * 4 + simd
    uniq, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
    per = cnt[inv]
    q = lambda a: '%.2f / %.2f / %.2f' % (np.quantile(a, 0.05), np.median(a), np.quantile(a, 0.95))
    print('rep %d: %d waves on %d SIMDs (%s with 1, %s with 2, %s with 3+); span %.2f us' % (
        rep, nw, len(uniq), (cnt == 1).sum(), (cnt == 2).sum(), (cnt > 2).sum(), us[:, 3].max()))
    for name, sel in (('all', per > 0), ('alone on SIMD', per == 1), ('two per SIMD', per == 2)):
        if not sel.any():
            continue
        u = us[sel]
        print('  %-14s start %s | loaded %s | issued %s | done %s   (5 %% / median / 95 %%)' % (name, q(u[:, 0]), q(u[:, 1]), q(u[:, 2]), q(u[:, 3])))
        print('  %-14s load phase %s | compute+store issue %s | drain %s' % ('', q(u[:, 1] - u[:, 0]), q(u[:, 2] - u[:, 1]), q(u[:, 3] - u[:, 2])))
    print('  waves per XCC:', np.bincount(xcc.astype(int)).tolist())
ms, b_alg, _ = mt.measure(warmup=5, iters=60)
print('with stamps: %.2f us per launch' % (ms * 1e3))
