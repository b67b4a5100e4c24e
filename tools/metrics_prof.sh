#!/bin/bash
# per-kernel durations of the error-statistics kernels (rocprofv3 kernel trace)
set -e
export TMPDIR=/tmp
rm -rf gpurun_out/prof_metrics
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_metrics -o m -- python3 tools/metrics_time.py > gpurun_out/metrics_time.log 2>&1
cat gpurun_out/metrics_time.log | tail -5
f=$(find gpurun_out/prof_metrics -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if 'sums' in r['Name'] or 'reduce_partials' in r['Name']:
        print(r['Name'][:70], 'calls', r['Calls'], 'avg_ns', r['AverageNs'], 'min', r['MinNs'], 'max', r['MaxNs'])
PY
f2=$(find gpurun_out/prof_metrics -name "*kernel_trace.csv" | head -1)
python3 - "$f2" <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'sums' in n:
        d[(n[:60], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', ''), r.get('Grid_Size_Y', ''))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in d.items():
    print(k, 'n', len(v), 'median_us', sorted(v)[len(v) // 2] / 1e3)
PY
