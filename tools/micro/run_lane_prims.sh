#!/bin/bash
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 lane_prims.hip -o /tmp/lane_prims
timeout -k 10 120 /tmp/lane_prims
