#!/bin/bash
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 exec_mask.hip -o /tmp/exec_mask
/tmp/exec_mask
