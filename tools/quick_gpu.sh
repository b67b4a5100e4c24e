#!/bin/bash
# tests + bench (no CPU baseline) in one GPU call; prints the headline numbers
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --timeout 300 > gpurun_out/pytest_q.log 2>&1; rc=$?; tail -3 gpurun_out/pytest_q.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py > gpurun_out/bench_q.json 2> gpurun_out/bench_q.err || { tail -5 gpurun_out/bench_q.err; exit 1; }
python - <<'PY'
import json
d = json.load(open('gpurun_out/bench_q.json'))
print('UNGM fused: %.3e steps/s  %.1f us/pass  frac %.4f' % (d['value'], 1e3 * d['ms_per_step'], d['roofline']['frac']))
m = d['roofline_mt6']
print('mt6: %.2f us  %.1f GB/s  frac %.4f  err %.2e' % (1e3 * m['ms_per_launch'], m['achieved'], m['frac'], m['max_scaled_err_vs_oracle']))
c = d['roofline_c5']
print('c5 gemm: %.1f us  %.1f TFLOP/s  frac %.3f  err %.1e  | whole D=10 N=201 transform B=1e4: %.2f ms' % (1e3 * c['ms_per_launch'], c['achieved'], c['frac'], c['max_scaled_err_vs_numpy'], c['full_transform_ms']))
PY
