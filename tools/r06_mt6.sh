#!/bin/bash
# round 6, the north-star transform: SSMQ_OPT_SYM against the LDL' kernel, partial-wave launches (library variants), per-wave timeline
export TMPDIR=/tmp
out=gpurun_out/r06_mt6.txt
: > $out
for r in 1 2; do
  for b in 65536 100000 131072 1000000; do
    MT6_B=$b MT6_ROUNDS=5 timeout -k 10 100 python tools/mt6_time.py | sed -e "s/^/sym   B=$b  /" >> $out
    SSMQ_NO_SYM=1 MT6_B=$b MT6_ROUNDS=5 timeout -k 10 100 python tools/mt6_time.py | sed -e "s/^/ldl   B=$b  /" >> $out
  done
done
for v in lpw48 lpw49 lpw56; do
  for e in "" "SSMQ_NO_SYM=1"; do
    env $e MT6_B=100000 MT6_ROUNDS=5 MT6_NOCHECK=1 SSMQ_LIBRARY=variants/libssmq_$v.so timeout -k 10 100 python tools/mt6_time.py | sed -e "s/^/$v $e  /" >> $out
  done
done
SSMQ_LIBRARY=variants/libssmq_stamp.so timeout -k 10 100 python tools/mt6_timeline.py > gpurun_out/r06_mt6_timeline_sym.txt 2>&1
SSMQ_NO_SYM=1 SSMQ_LIBRARY=variants/libssmq_stamp.so timeout -k 10 100 python tools/mt6_timeline.py > gpurun_out/r06_mt6_timeline_ldl.txt 2>&1
MT6_B=65536 SSMQ_LIBRARY=variants/libssmq_stamp.so timeout -k 10 100 python tools/mt6_timeline.py > gpurun_out/r06_mt6_timeline_sym_65536.txt 2>&1
cat $out
