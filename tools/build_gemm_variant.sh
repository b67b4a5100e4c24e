#!/bin/bash
# tools/build_gemm_variant.sh NAME "EXTRA_FLAGS": variants/libssmq_NAME.so = the current objects with ssmq_gemm_mfma.hip
# rebuilt under EXTRA_FLAGS, for A/B timing with tools/c5_ab.py (SSMQ_LIBRARY=variants/libssmq_NAME.so).
set -e
cd "$(dirname "$0")/../ssmtoybox_amd/csrc"
mkdir -p ../../variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $2 -c ssmq_gemm_mfma.hip -o ../../variants/gemm_$1.o
objs=$(ls *.o | grep -v ssmq_gemm_mfma.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libssmq_$1.so $objs ../../variants/gemm_$1.o -ldl -lpthread
echo built variants/libssmq_$1.so
