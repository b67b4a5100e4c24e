#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/bench_r06_detail.json > gpurun_out/bench_r06.out 2> gpurun_out/bench_r06.err || { tail -20 gpurun_out/bench_r06.err; exit 1; }
python3 - <<'PY'
import json
lines = open('gpurun_out/bench_r06.out').read().splitlines()
print('stdout lines:', len(lines), ' last line bytes:', len(lines[-1]))
d = json.loads(lines[-1])
print(lines[-1])
PY
