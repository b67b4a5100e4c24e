#!/bin/bash
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --timeout 300 2>&1 | tail -3 || exit 1
for r in 1 2 3; do
  SSMQ_NO_FASTPATH=1 SSMQ_LIBRARY=variants/libssmq_fast.so python tools/mt6_time.py | sed 's/^/dense  /'
  SSMQ_LIBRARY=variants/libssmq_fast.so python tools/mt6_time.py | sed 's/^/fast   /'
done
