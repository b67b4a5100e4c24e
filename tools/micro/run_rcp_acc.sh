#!/bin/bash
set -e
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o /tmp/rcp_acc tools/micro/rcp_acc.hip
timeout -k 10 120 /tmp/rcp_acc
