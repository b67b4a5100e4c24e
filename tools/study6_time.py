#!/usr/bin/env python3
"""Six configs[1]-sized filters (UKF, CKF, GHKF, GPQKF, TPQKF, BSQKF on the same UNGM measurements; research/bsq/bsq_ungm.py:132-137,
research/tpq/tpq_base.py:175-192) device-resident: one after the other on one stream, as one launch graph
(ssmq_filter_forward_multi_dev), and the branches without the graph (SSMQ_MULTI_NO_GRAPH=1).  HIP events, median of blocks."""
import ctypes
import os
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssmtoybox_amd as amd  # noqa: E402
from benchlib.legs import Study6Bench  # noqa: E402

amd.set_device(0)
B, T = int(os.environ.get('B', '10000')), int(os.environ.get('T', '100'))
st = Study6Bench(amd, B, T, seed=1)
print('filters:', ', '.join(st.names))
one = st.time_single(0)
ser = st.time_serial()
os.environ.pop('SSMQ_MULTI_NO_GRAPH', None)
mul = st.time_multi()
os.environ['SSMQ_MULTI_NO_FAMILY'] = '1'
mul_g = st.time_multi()
os.environ['SSMQ_MULTI_NO_GRAPH'] = '1'
mul_ng = st.time_multi()
os.environ.pop('SSMQ_MULTI_NO_GRAPH', None)
os.environ.pop('SSMQ_MULTI_NO_FAMILY', None)
print('one pass (GPQKF)                 %.1f us' % (one * 1e3))
print('six passes, one after the other  %.1f us  (%.2f x one pass)' % (ser * 1e3, ser / one))
print('six passes, ONE kernel (family)   %.1f us  (%.2f x one pass)  -> %.3e filter steps/s' % (mul * 1e3, mul / one, 6 * B * T / (mul * 1e-3)))
print('six passes, one forked graph     %.1f us  (%.2f x one pass)' % (mul_g * 1e3, mul_g / one))
print('six passes, branches, no graph   %.1f us  (%.2f x one pass)' % (mul_ng * 1e3, mul_ng / one))
print('results equal to the serial calls:', st.check())
st.free()
