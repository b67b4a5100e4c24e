#!/bin/bash
# Round-3 GPU check: parity suite, default bench line, the self-spawned 2-rank launch (weak + strong split).
set -e
export TMPDIR=/tmp
out=gpurun_out/${1:-r03_check}
mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
python bench.py --gpus 2 --steps 20 --warmup 5 > $out/bench_gpus2.json 2> $out/bench_gpus2.err || { tail -20 $out/bench_gpus2.err; exit 1; }
python bench.py --gpus 2 --steps 10 --warmup 3 --workload reentry6 --filter ukf --total-batch 100000 --time-steps 50 > $out/bench_gpus2_strong.json 2> $out/bench_gpus2_strong.err || { tail -20 $out/bench_gpus2_strong.err; exit 1; }
python - $out <<'PY'
import json, sys
o = sys.argv[1]
b = json.load(open(o + '/bench.json'))
print('N=1 value %.3e ms %.4f roofline %.3f target %s' % (b['value'], b['ms_per_step'], b['roofline']['frac'], b['roofline'].get('target', {}).get('frac')))
print('saturated', [(r['mc'], round(r['ms_per_launch'], 4), '%.2e' % r['filter_steps_per_s'], round(r['frac'], 3), round(r.get('issue_frac_chip', 0), 3)) for r in b['roofline'].get('saturated', [])])
for f in ('bench_gpus2', 'bench_gpus2_strong'):
    b = json.load(open(o + '/' + f + '.json'))
    print(f, 'n_gpus', b['n_gpus'], 'scaling', b['scaling'], 'value %.3e' % b['value'], b['config'].get('per_rank_kernel_ms'), b['config'].get('per_rank_trajectories'), 'allreduce_us', b['config'].get('allreduce_us'), b['config']['collective'], b.get('trajectories_aggregated'))
PY
