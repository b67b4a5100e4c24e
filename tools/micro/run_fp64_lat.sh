#!/bin/bash
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 fp64_lat.hip -o /tmp/fp64_lat
/tmp/fp64_lat
