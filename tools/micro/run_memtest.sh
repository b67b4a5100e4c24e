#!/bin/bash
set -e
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/memtest tools/micro/memtest.hip
timeout -k 10 200 /tmp/memtest
