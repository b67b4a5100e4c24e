#!/bin/bash
# MFMA busy cycles of the matrix-core GEMM (separate PMC pass, no tracing domains): SQ_VALU_MFMA_BUSY_CYCLES vs the
# kernel's duration gives the utilisation of the matrix pipe; cycles per MFMA = busy cycles / instruction count.
export TMPDIR=/tmp
out=gpurun_out/pmc_mfma
rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/a -- python3 tools/c5_ab.py > $out/a.log 2>&1 || { tail -5 $out/a.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA --output-format csv -d $out/b -- python3 tools/c5_ab.py > $out/b.log 2>&1 || tail -3 $out/b.log
python3 - <<'PY'
import csv, glob, collections
for d in ('a', 'b'):
    for f in glob.glob('gpurun_out/pmc_mfma/%s/**/*counter_collection.csv' % d, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'fxwc' in r['Kernel_Name']:
                acc[(r['Counter_Name'], r['Grid_Size'])].append(float(r['Counter_Value']))
        for k, v in sorted(acc.items()):
            print(d, k, 'n', len(v), 'mean', sum(v) / len(v))
PY
