#!/bin/bash
# timing experiments on k_fxwc_mfma: library variants with the barrier and / or the slab traffic compiled out (results wrong)
export TMPDIR=/tmp
echo base; timeout -k 10 120 python tools/c5_ab.py
for lib in variants/libssmq_*.so; do echo $lib; SSMQ_LIBRARY=$lib timeout -k 10 120 python tools/c5_ab.py 2>&1 | grep -v Assert | tail -3; done
