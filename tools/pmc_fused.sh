#!/bin/bash
# SQ counters of the bench kernels (k_filter_fused on UNGM - the headline - and on the reentry / coordinated-turn models,
# the D = 6 transform, the D = 10 passes): two PMC passes over bench.py, no tracing domains.
# Output: gpurun_out/pmc_fused/summary.csv (copied to profiles/rNN_fused_sq.csv).
set -e
out=${1:-gpurun_out/pmc_fused}
export TMPDIR=/tmp
mkdir -p "$out"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$out/a" -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > "$out/a.log" 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d "$out/b" -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > "$out/b.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
want = ('k_filter_fused', 'k_filter_chunked', 'k_filter_wsplit', 'k_apply_small<6', 'k_fxwc', 'k_eval_wave', 'k_apply_tile', 'k_apply_wave', 'k_big_rest', 'k_bq_fused', 'k_bq_stream')
for sub in ('a', 'b'):
    for path in glob.glob(root + '/' + sub + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(path)):
            n = r['Kernel_Name']
            if any(w in n.replace(' ', '') for w in want):
                acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
with open(root + '/summary.csv', 'w') as f:
    f.write('kernel,counter,mean_per_launch,launches\n')
    for name, d in sorted(acc.items()):
        for c, v in sorted(d.items()):
            f.write('"%s",%s,%.1f,%d\n' % (name, c, sum(v) / len(v), len(v)))
print(open(root + '/summary.csv').read())
PY
