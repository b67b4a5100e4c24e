#!/bin/bash
# HBM traffic of k_apply_tile at D = E = 10, N = 21, B = 1e5 (tools/c5_n21.py): FETCH_SIZE / WRITE_SIZE in separate passes
export TMPDIR=/tmp
out=gpurun_out/pmc_tile
rm -rf $out; mkdir -p $out
python3 tools/c5_n21.py 100000 > $out/time.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 tools/c5_n21.py 100000 > $out/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 tools/c5_n21.py 100000 > $out/w.log 2>&1
cat $out/time.log
python3 - $out <<'PY'
import csv, glob, sys, collections
for d in ('fetch', 'write'):
    acc = collections.defaultdict(list)
    for path in glob.glob(sys.argv[1] + '/' + d + '/**/*counter_collection.csv', recursive=True):
        by = collections.defaultdict(float); nm = {}
        for r in csv.DictReader(open(path)):
            by[r['Dispatch_Id']] += float(r['Counter_Value']); nm[r['Dispatch_Id']] = r['Kernel_Name'][:50]
        for k, v in by.items(): acc[nm[k]].append(v)
    for k, v in sorted(acc.items()):
        if 'tile' in k or 'aos' in k: print(d, k, len(v), 'mean KiB', sum(v) / len(v))
PY
