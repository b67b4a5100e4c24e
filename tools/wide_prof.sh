#!/bin/bash
set -e
export TMPDIR=/tmp
rm -rf gpurun_out/prof_wide
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_wide -o w -- python3 tools/wide_time.py ${1:-2048} > gpurun_out/wide_time.log 2>&1
grep "wall" gpurun_out/wide_time.log
python3 - <<'PY'
import csv, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/prof_wide/w_kernel_trace.csv')):
    d[r['Kernel_Name'][:50]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in d.items():
    print(k, 'n', len(v), 'us', [round(x / 1e3, 1) for x in v][:8])
PY
