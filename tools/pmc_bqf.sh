#!/bin/bash
# SQ counters of k_bq_fused (and of the two-pass kernels with SSMQ_NO_BQ_FUSED=1): separate PMC passes, no tracing domains.
export TMPDIR=/tmp
out=gpurun_out/pmc_bqf
rm -rf $out; mkdir -p $out
export C5_B=10000
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $out/a -- python3 tools/c5_full.py > $out/a.log 2>&1 || { tail -5 $out/a.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $out/b -- python3 tools/c5_full.py > $out/b.log 2>&1 || tail -3 $out/b.log
rocprofv3 --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --output-format csv -d $out/c -- python3 tools/c5_full.py > $out/c.log 2>&1 || tail -3 $out/c.log
python3 - <<'PY'
import csv, glob, collections
for d in ('a', 'b', 'c'):
    for f in glob.glob('gpurun_out/pmc_bqf/%s/**/*counter_collection.csv' % d, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'bq_fused' in r['Kernel_Name'] or 'fxwc' in r['Kernel_Name'] or 'eval_wave' in r['Kernel_Name']:
                acc[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k, v in sorted(acc.items()):
            print(d, k, 'n', len(v), 'mean %.4g' % (sum(v) / len(v)))
PY
