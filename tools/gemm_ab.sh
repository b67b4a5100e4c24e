#!/bin/bash
# A/B of the row-block size of the matrix-core GEMM at a few batch sizes (kernel time from rocprofv3)
export TMPDIR=/tmp
for rt in 1 2; do for B in 1024 4096; do
  rm -rf gpurun_out/prof_wide
  SSMQ_GEMM_RT=$rt rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_wide -o w -- python3 tools/wide_time.py $B > gpurun_out/wide_time.log 2>&1 || { tail -5 gpurun_out/wide_time.log; exit 1; }
  python3 - $rt $B <<'PY'
import csv, sys
for r in csv.DictReader(open('gpurun_out/prof_wide/w_kernel_trace.csv')):
    if 'fxwc' in r['Kernel_Name']:
        print('RT', sys.argv[1], 'B', sys.argv[2], 'us', (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
PY
done; done
