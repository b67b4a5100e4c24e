#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out

timeout -k 10 600 python -m pytest tests -q -m gpu --timeout 300 > gpurun_out/pytest_r06b.log 2>&1; tail -8 gpurun_out/pytest_r06b.log
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_r06b.log | head
timeout -k 10 200 python tools/api_rate.py > gpurun_out/r06_api_rate.txt 2>&1; cat gpurun_out/r06_api_rate.txt
timeout -k 10 200 python tools/study6_time.py > gpurun_out/r06_study6.txt 2>&1; cat gpurun_out/r06_study6.txt
SSMQ_LIBRARY=variants/libssmq_stamp.so timeout -k 10 100 python tools/mt6_timeline.py > gpurun_out/r06_mt6_timeline_sym.txt 2>&1; head -12 gpurun_out/r06_mt6_timeline_sym.txt
