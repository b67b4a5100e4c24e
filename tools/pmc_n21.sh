#!/bin/bash
# SQ counters of the D = 10, N = 21 transform (k_apply_wave): two PMC passes, no tracing
export TMPDIR=/tmp
out=gpurun_out/pmc_n21
rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $out/a -- python3 tools/c5_n21.py 100000 > $out/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d $out/b -- python3 tools/c5_n21.py 100000 > $out/b.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for path in glob.glob(sys.argv[1] + '/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        if 'k_apply_wa' in r['Kernel_Name'] or 'k_apply_wide' in r['Kernel_Name'] or 'k_apply_tile' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print('%-24s %14.1f  (%d launches)' % (k, sum(v) / len(v), len(v)))
PY
