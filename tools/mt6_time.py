#!/usr/bin/env python3
"""Time the D = E = 6, N = 13 moment-transform kernel (bench.py's Mt6Bench) in isolation: several event-timed batches,
prints min / median per-launch time.  Used for A/B runs of library variants (SSMQ_LIBRARY=path/to/libssmq_x.so)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from bench import Mt6Bench  # noqa: E402

amd.set_device(0)
mt = Mt6Bench(amd, int(os.environ.get('MT6_B', '100000')), seed=2)
err = mt.check() if not os.environ.get("MT6_NOCHECK") else float("nan")
times = []
for r in range(int(os.environ.get('MT6_ROUNDS', '7'))):
    ms, b_alg, _ = mt.measure(warmup=5, iters=60)
    times.append(ms * 1e3)
print('%-28s min %.2f us  median %.2f us  -> %.0f GB/s (median)  err %.1e' % (
    os.path.basename(os.environ.get('SSMQ_LIBRARY', 'libssmq.so')), min(times), statistics.median(times),
    b_alg / statistics.median(times) / 1e3, err))
