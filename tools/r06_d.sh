#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -q -m gpu --timeout 300 > gpurun_out/pytest_r06d.log 2>&1; tail -6 gpurun_out/pytest_r06d.log
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_r06d.log | head
: > gpurun_out/r06_sym_filters.txt
for cfg in "reentry5 bsqkf 100000 50" "reentry5 bsqkf 12500 50" "reentry6 bsqkf 100000 50" "reentry5 gpqkf 100000 50" "reentry6 gpqkf 100000 50"; do
  set -- $cfg
  for e in "" "SSMQ_NO_SYM=1"; do
    env $e WL=$1 FILT=$2 B=$3 T=$4 timeout -k 10 120 python tools/fused_time.py 2>&1 | tail -1 | sed -e "s/^/$1 $2 B=$3 T=$4 ${e:-sym}: /" >> gpurun_out/r06_sym_filters.txt
  done
done
cat gpurun_out/r06_sym_filters.txt
