#!/usr/bin/env python3
"""Whole D = E = 10, N = 201 Bayes-Sard transform at B = 1e4 ... 4e4 (bench.py: C5GemmBench.measure_full_transform), best
of five - for A/B runs of the one-launch route (k_bq_fused) against the two-pass one (SSMQ_NO_BQ_FUSED=1)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from bench import C5GemmBench  # noqa: E402

amd.set_device(0)
for B in [int(v) for v in os.environ.get("C5_B", "10000,40000").split(",")]:
    c5 = C5GemmBench(amd, B, seed=5)
    ts = [c5.measure_full_transform(B, with_cpu=False)[0] for _ in range(5)]
    print('%s B=%6d  best %.1f us  median %.1f us' % ('two-pass' if os.environ.get('SSMQ_NO_BQ_FUSED') else 'one-launch',
                                                     B, 1e3 * min(ts), 1e3 * sorted(ts)[2]), flush=True)
