#!/usr/bin/env python3
"""A/B of library variants on the C5 matrix-core GEMM (SSMQ_LIBRARY=...)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from bench import C5GemmBench  # noqa: E402

amd.set_device(0)
for B in (2048, 10000, 40000):
    c5 = C5GemmBench(amd, B, seed=5)
    err = c5.check()
    ts = [c5.measure()[0] for _ in range(5)]
    flop = 2.0 * c5.M * c5.NP * c5.NP
    print('%-22s B=%6d  %.1f us  %.1f TFLOP/s  err %.1e' % (os.path.basename(os.environ.get('SSMQ_LIBRARY', 'libssmq.so')),
                                                          B, 1e3 * min(ts), flop / (min(ts) * 1e-3) / 1e12, err))
