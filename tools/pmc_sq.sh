#!/bin/bash
# SQ-level counters for the bench kernels (separate PMC pass, no tracing domains).
set -e
out=${1:-gpurun_out/pmc_sq}
export TMPDIR=/tmp
mkdir -p "$out"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$out/a" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$out/a.log" 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d "$out/b" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$out/b.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
for sub in ('a', 'b'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(root + '/' + sub + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(path)):
            n = r['Kernel_Name']
            if 'k_apply_small<6' in n.replace(' ', '') or 'k_filter_fused' in n:
                acc[n.split('(')[0][-60:]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        print(k)
        for c, v in sorted(d.items()):
            print('   %-24s %14.1f  (n=%d)' % (c, sum(v) / len(v), len(v)))
PY
