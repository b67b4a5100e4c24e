#!/bin/bash
# kernel times of the error-statistics kernels for library variants (rocprofv3 kernel trace)
export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf gpurun_out/prof_metrics
  SSMQ_LIBRARY=$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_metrics -o m -- python3 tools/metrics_time.py > gpurun_out/metrics_time.log 2>&1 || { tail -5 gpurun_out/metrics_time.log; exit 1; }
  python3 - $lib <<'PY'
import csv, sys
for r in csv.DictReader(open('gpurun_out/prof_metrics/m_kernel_stats.csv')):
    if 'sums<' in r['Name']:
        print(sys.argv[1].split('/')[-1], r['Name'].split('::')[-1][:18], 'avg_us', round(float(r['AverageNs']) / 1e3, 1))
PY
done
