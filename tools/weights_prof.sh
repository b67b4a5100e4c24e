#!/bin/bash
# kernel trace of the weights kernels at a few shapes (tools/weights_time.py)
export TMPDIR=/tmp
rm -rf gpurun_out/prof_weights
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_weights -o w -- python3 tools/weights_time.py > gpurun_out/weights_time.log 2>&1 || { tail -5 gpurun_out/weights_time.log; exit 1; }
cat gpurun_out/weights_time.log | tail -8
python3 - <<'PY'
import csv, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/prof_weights/w_kernel_trace.csv')):
    if 'k_weights' in r['Kernel_Name']:
        g = int(r.get('Grid_Size_X', r.get('Grid_Size', 0)))
        d[(r['Kernel_Name'][:60], g, int(r.get('Workgroup_Size_X', 0)), int(r.get('LDS_Block_Size', r.get('LDS_Block_Size_v', 0)) or 0))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items()):
    print('%-62s grid %7d wg %4d lds %6d  n %3d  median %9.1f us' % (k + (len(v), sorted(v)[len(v) // 2])))
PY
