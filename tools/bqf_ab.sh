#!/bin/bash
# timing experiments on k_bq_fused: library variants with one phase compiled out (results wrong, durations telling)
export TMPDIR=/tmp
echo base; C5_B=10000 timeout -k 10 120 python tools/c5_full.py
for lib in variants/libssmq_*.so; do echo $lib; SSMQ_LIBRARY=$lib C5_B=10000 timeout -k 10 120 python tools/c5_full.py; done
