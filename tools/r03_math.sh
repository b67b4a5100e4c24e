#!/bin/bash
# after the elementary-function sequences: parity suite, bench line (fused reentry filters), tile kernel timing
export TMPDIR=/tmp
out=gpurun_out/${1:-r03_math}
mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $out/pytest.log
cp gpurun_out/parity_stats.json $out/parity_stats.json 2>/dev/null
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err || tail -5 $out/bench.err
python - $out <<'PY'
import json, sys
b = json.load(open(sys.argv[1] + '/bench.json'))
print('headline %.3e  %.4f ms' % (b['value'], b['ms_per_step']))
for k, v in b['roofline_c3'].items():
    print(k, '%.3f ms' % v['ms_per_launch'], 'frac %.3f' % v['frac'], v.get('mean_diff_vs_cpu_port_in_sigmas'))
v = b['roofline_c4']; print('c4 %.3f ms' % v['ms_per_launch'], v.get('mean_diff_vs_cpu_port_in_sigmas'))
v = b['roofline_c5']; print('c5 gemm %.3f ms full %.3f ms n21 %.3f ms' % (v['ms_per_launch'], v['full_transform_ms'], v['unisolvent_n21']['ms_per_launch']), v['unisolvent_n21']['kernel'])
print('mt6 %.4f ms frac %.3f' % (b['roofline_mt6']['ms_per_launch'], b['roofline_mt6']['frac']))
PY
