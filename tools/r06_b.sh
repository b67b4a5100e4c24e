#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 200 python tools/chunk_debug.py > gpurun_out/r06_chunk_debug.txt 2>&1; cat gpurun_out/r06_chunk_debug.txt
timeout -k 10 600 python -m pytest tests -q -m gpu --timeout 300 > gpurun_out/pytest_r06b.log 2>&1; tail -8 gpurun_out/pytest_r06b.log
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_r06b.log | head
timeout -k 10 200 python tools/api_rate.py > gpurun_out/r06_api_rate.txt 2>&1; cat gpurun_out/r06_api_rate.txt
timeout -k 10 200 python tools/study6_time.py > gpurun_out/r06_study6.txt 2>&1; cat gpurun_out/r06_study6.txt
bash tools/r06_mt6.sh > /dev/null 2>&1; cat gpurun_out/r06_mt6.txt | grep -v lpw; head -12 gpurun_out/r06_mt6_timeline_sym.txt
