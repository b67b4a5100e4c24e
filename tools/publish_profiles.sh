#!/bin/bash
# copies the summaries of a tools/final_profiles.sh run (gpurun_out/<tag>/) into profiles/ under the names profiles/README.md cites
set -e
tag=${1:?tag, e.g. r05_b}
src=gpurun_out/$tag
tail -1 $src/bench.json > profiles/${tag}_bench.json
cp $src/bench_detail.json profiles/${tag}_bench_detail.json
cp $src/kernel_stats.csv profiles/${tag}_bench_kernel_stats.csv
cp $src/kernel_stats_by_grid.csv profiles/${tag}_bench_kernel_stats_by_grid.csv
cp $src/pmc_traffic.json profiles/pmc_traffic.json
rnd=${tag%%_*}
cp $src/pmc_fused/summary.csv profiles/${rnd}_fused_sq.csv
cp $src/perf_gate.md profiles/${tag}_perf_gate.md
[ -f gpurun_out/parity_stats.json ] && python3 tools/keep_larger_json.py gpurun_out/parity_stats.json profiles/${rnd}_parity_stats.json
ls -la profiles/${tag}_* profiles/pmc_traffic.json profiles/${rnd}_fused_sq.csv
