#!/bin/bash
# tests + bench in one GPU call; prints the result line (what the driver keeps) and its length
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -q -m gpu -x --timeout 300 > gpurun_out/pytest_q.log 2>&1; rc=$?; tail -3 gpurun_out/pytest_q.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)|Error" gpurun_out/pytest_q.log | head -20; exit $rc; }
# the driver's command
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/bench_q_detail.json > gpurun_out/bench_q.out 2> gpurun_out/bench_q.err || { tail -5 gpurun_out/bench_q.err; exit 1; }
python3 - <<'PY'
import json
lines = open('gpurun_out/bench_q.out').read().splitlines()
print('stdout lines:', len(lines), ' last line bytes:', len(lines[-1]))
d = json.loads(lines[-1])
print(lines[-1])
print('UNGM fused: %.3e steps/s  %.1f us/pass (median of blocks %.1f)  frac %.4f | target frac %.4f | cpu %s' % (
    d['value'], 1e3 * d['ms_per_step'], 1e3 * d.get('ms_per_step_median', 0), d['roofline']['frac'], d['roofline'].get('target_frac', -1),
    d.get('cpu_baseline', {}).get('cpu_model')))
PY
