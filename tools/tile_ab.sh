#!/bin/bash
# k_apply_tile variants (tools/tile_variants.sh) on the D = E = 10, N = 21, B = 1e5 transform: ms per launch, three readings each
export TMPDIR=/tmp
echo base; timeout -k 10 120 python tools/n21_bench.py
for lib in variants/libssmq_tile_*.so; do echo $lib; SSMQ_LIBRARY=$lib timeout -k 10 120 python tools/n21_bench.py; done
