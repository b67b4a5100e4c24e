#!/bin/bash
# Builds library variants that differ in the compile-time switches of k_apply_tile (variants/libssmq_tile_<name>.so; run them
# with SSMQ_LIBRARY=... tools/n21_bench.py).  usage: tools/tile_variants.sh name "-DSSMQ_TILE_OCC=3 ..." [name flags ...]
set -e
cd "$(dirname "$0")/../ssmtoybox_amd/csrc"
mkdir -p ../../variants
others=$(ls *.o | grep -v '^ssmq_apply_tile.o$')
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-result -Wno-unused-value $flags -c ssmq_apply_tile.hip -o ../../variants/tile_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libssmq_tile_$name.so ../../variants/tile_$name.o $others -ldl -lpthread
  echo "built variants/libssmq_tile_$name.so ($flags)"
done
