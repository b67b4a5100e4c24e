#!/bin/bash
# SQ counters of the headline kernel (k_filter_fused, UNGM GPQ-Kalman B=1e4 x T=100): two PMC passes, no tracing domains.
# Output: gpurun_out/pmc_fused/summary.csv (copied to profiles/r02_fused_sq.csv).
set -e
out=${1:-gpurun_out/pmc_fused}
export TMPDIR=/tmp
mkdir -p "$out"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$out/a" -- python3 tools/fused_time.py > "$out/a.log" 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d "$out/b" -- python3 tools/fused_time.py > "$out/b.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(list)
name = None
for sub in ('a', 'b'):
    for path in glob.glob(root + '/' + sub + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(path)):
            if 'k_filter_fused' in r['Kernel_Name']:
                name = r['Kernel_Name']
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
with open(root + '/summary.csv', 'w') as f:
    f.write('kernel,counter,mean_per_launch,launches\n')
    for c, v in sorted(acc.items()):
        f.write('"%s",%s,%.1f,%d\n' % (name, c, sum(v) / len(v), len(v)))
print(open(root + '/summary.csv').read())
PY
