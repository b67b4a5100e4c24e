#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -q -m gpu --timeout 300 > gpurun_out/pytest_r06g.log 2>&1; tail -4 gpurun_out/pytest_r06g.log
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_r06g.log | head
timeout -k 10 300 python tools/quad_time.py > gpurun_out/r06_quad2.txt 2>&1; cat gpurun_out/r06_quad2.txt
timeout -k 10 200 python tools/api_rate.py 2>&1 | head -3
