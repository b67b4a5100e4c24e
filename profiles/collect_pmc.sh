#!/bin/bash
# HBM traffic counters for the bench kernels, collected as the MI355X guide prescribes: PMC passes on their own (no
# tracing domains), FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they do not fit one pass on gfx950).  Run on the GPU box
# from the repo root:  bash profiles/collect_pmc.sh gpurun_out/pmc
# then summarise:      python profiles/pmc_summary.py gpurun_out/pmc > profiles/pmc_traffic.json
set -e
out=${1:-gpurun_out/pmc}
export TMPDIR=/tmp
mkdir -p "$out"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$out/write.log" 2>&1
find "$out" -name '*counter_collection.csv' | head
