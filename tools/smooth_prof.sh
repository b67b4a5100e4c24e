#!/bin/bash
export TMPDIR=/tmp
rm -rf gpurun_out/prof_smooth
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_smooth -o s -- python3 tools/smooth_prof.py > gpurun_out/smooth.log 2>&1 || { tail -5 gpurun_out/smooth.log; exit 1; }
tail -1 gpurun_out/smooth.log
head -6 gpurun_out/prof_smooth/s_kernel_stats.csv | cut -c1-150
