#!/bin/bash
# tools/build_file_variant.sh FILE NAME "EXTRA_FLAGS": variants/libssmq_NAME.so = the current objects with csrc/FILE.hip rebuilt
# under EXTRA_FLAGS.  For A/B timing (SSMQ_LIBRARY=variants/libssmq_NAME.so python tools/...).
set -e
cd "$(dirname "$0")/../ssmtoybox_amd/csrc"
mkdir -p ../../variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $3 -c $1.hip -o ../../variants/$1_$2.o
objs=$(ls *.o | grep -v "^$1.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libssmq_$2.so $objs ../../variants/$1_$2.o -ldl -lpthread
echo built variants/libssmq_$2.so
