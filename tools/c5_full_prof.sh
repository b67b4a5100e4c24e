#!/bin/bash
export TMPDIR=/tmp
rm -rf gpurun_out/prof_c5
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_c5 -o c -- python3 tools/c5_full.py ${1:-10000} > gpurun_out/c5_full.log 2>&1 || { tail -5 gpurun_out/c5_full.log; exit 1; }
grep "whole" gpurun_out/c5_full.log
python3 - <<'PY'
import csv, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/prof_c5/c_kernel_trace.csv')):
    if int(r['Grid_Size_X'] if 'Grid_Size_X' in r else r['Grid_Size']) >= 64 * 5000:
        d[r['Kernel_Name'][:48]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in d.items():
    print(k, 'n', len(v), 'median_us', sorted(v)[len(v) // 2], 'first', v[:4])
PY
