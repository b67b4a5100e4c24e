#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/mgprof
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/marginal_dev_run.py > $out/run.txt 2>&1
cat $out/run.txt | tail -3
f=$(find $out -name '*kernel_stats.csv' | head -1)
cut -c1-200 $f | head -14
