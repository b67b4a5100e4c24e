#!/bin/bash
# kernel trace of the theta-batched step at 65 536 items (UNGM and pendulum): where a call's device time goes
export TMPDIR=/tmp
rm -rf gpurun_out/prof_theta
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_theta -o t -- python3 tools/theta_time.py > gpurun_out/theta_prof_run.log 2>&1 || { tail -5 gpurun_out/theta_prof_run.log; exit 1; }
python3 - <<'PY'
import csv, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/prof_theta/t_kernel_trace.csv')):
    g = int(r['Grid_Size_X'] if 'Grid_Size_X' in r else r['Grid_Size'])
    if g >= 65536 * 8:
        d[(r['Kernel_Name'][:60], g)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print('%-62s grid %9d n %4d median_us %9.1f' % (k[0], k[1], len(v), sorted(v)[len(v) // 2]))
PY
