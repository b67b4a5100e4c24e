#!/bin/bash
export TMPDIR=/tmp
echo base; timeout -k 10 120 python tools/c5_n21.py 100000 | tail -1
for lib in variants/libssmq_*.so; do echo $lib; SSMQ_LIBRARY=$lib timeout -k 10 120 python tools/c5_n21.py 100000 | tail -1; done
