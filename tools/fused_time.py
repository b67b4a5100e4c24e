"""Kernel time of the fused GPQ-Kalman loop on UNGM (BASELINE configs[1]: B = 1e4, T = 100), HIP events around N
back-to-back launches.  SSMQ_LIBRARY selects a library variant, SSMQ_FUSED_QUAD=0/1 the kernel variant."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd                      # noqa: E402
from ssmtoybox_amd import _lib                   # noqa: E402
from bench import FilterBench                    # noqa: E402

B = int(os.environ.get('B', 10000))
T = int(os.environ.get('T', 100))
filt = os.environ.get('FILT', 'gpqkf')
N = int(os.environ.get('N', 100))
wl = FilterBench(amd, B, T, 1, os.environ.get('WL', 'ungm'), filt)
for _ in range(5):
    wl.step()
_lib.sync()
ts = []
for rep in range(5):
    e0, e1 = _lib.Event(), _lib.Event()
    e0.record()
    for _ in range(N):
        wl.step()
    e1.record()
    _lib.sync()
    ts.append(e0.elapsed_ms(e1) * 1e3 / N)          # us per launch
fm, fP, st = wl.results()
print('%-40s B=%d T=%d  %s  us/launch: median %.2f min %.2f  checksum %.17g' %
      (os.path.basename(os.environ.get('SSMQ_LIBRARY', 'libssmq.so')), B, T,
       wl.kernel[-40:], float(np.median(ts)), min(ts), float(np.nansum(fm) + np.nansum(fP))))
