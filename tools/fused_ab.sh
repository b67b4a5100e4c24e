#!/bin/bash
# A/B of fused-kernel library variants (tools/build_variant.sh): tools/fused_ab.sh variants/libssmq_a.so ...
for lib in "$@"; do
  SSMQ_LIBRARY=$lib python tools/fused_time.py
done
