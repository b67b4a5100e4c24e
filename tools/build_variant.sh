#!/bin/bash
# tools/build_variant.sh NAME "EXTRA_FLAGS": variants/libssmq_NAME.so = the current objects with ssmq_filter_fused.hip
# rebuilt (UNGM kernels only, for speed; SSMQ_VARIANT_SCOPE= for all of them) under EXTRA_FLAGS.  For A/B timing with tools/fused_time.py.
set -e
cd "$(dirname "$0")/../ssmtoybox_amd/csrc"
mkdir -p ../../variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 ${SSMQ_VARIANT_SCOPE--DSSMQ_FUSED_UNGM_ONLY} $2 -c ssmq_filter_fused.hip -o ../../variants/fused_$1.o
objs=$(ls *.o | grep -v ssmq_filter_fused.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libssmq_$1.so $objs ../../variants/fused_$1.o
echo built variants/libssmq_$1.so
