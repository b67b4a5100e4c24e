#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -q -m gpu --timeout 300 > gpurun_out/pytest_r06e.log 2>&1; tail -6 gpurun_out/pytest_r06e.log
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_r06e.log | head
for e in "" "SSMQ_NO_SYM=1"; do env $e timeout -k 10 120 python tools/fused_time.py 2>&1 | tail -1 | sed -e "s/^/ungm gpqkf ${e:-sym}: /"; done
bash tools/r06_bench.sh
