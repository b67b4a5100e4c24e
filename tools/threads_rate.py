#!/usr/bin/env python3
"""Aggregate rate of K threads that each run their own filter (configs[1]: GPQ-Kalman on UNGM, B = 1e4, T = 100, device-resident
passes) - every thread has its own stream and workspace (include/ssmq.h, "Threads"), so the passes of different threads overlap
on the device: one pass fills 157 of the chip's 1 024 SIMDs."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from ssmtoybox_amd import _lib  # noqa: E402
from benchlib.workloads import FilterBench  # noqa: E402

amd.set_device(0)
B, T, passes = 10000, 100, 400


def worker(k, out, barrier):
    wl = FilterBench(amd, B, T, 100 + k)
    for _ in range(20):
        wl.step()
    _lib.sync()
    barrier.wait()
    t0 = time.perf_counter()
    for _ in range(passes):
        wl.step()
    _lib.sync()
    out[k] = (t0, time.perf_counter())


for K in [int(v) for v in sys.argv[1:]] or [1, 2, 4, 6, 8]:
    out, barrier = {}, threading.Barrier(K)
    th = [threading.Thread(target=worker, args=(k, out, barrier)) for k in range(K)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    t0, t1 = min(v[0] for v in out.values()), max(v[1] for v in out.values())
    print('%d thread(s): %.1f us per pass and thread, %.3e filter steps/s in aggregate' % (
        K, 1e6 * (t1 - t0) / passes, K * passes * B * T / (t1 - t0)), flush=True)
