#!/usr/bin/env python3
"""The whole Bayes-Sard D = 10, N = 201 transform on device-resident moments (three launches): run under rocprofv3
--kernel-trace for the per-pass times."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from bench import C5GemmBench  # noqa: E402

amd.set_device(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
c5 = C5GemmBench(amd, 64, seed=5)
print('B=%d whole transform %.3f ms' % (B, c5.measure_full_transform(B, False)[0]))
