#!/bin/bash
# Produces the artefacts committed under profiles/: bench JSON, rocprofv3 kernel stats of the same command, PMC traffic.
set -e
export TMPDIR=/tmp
tag=${1:-r02_final}
out=gpurun_out/$tag
mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --no-cpu-baseline > $out/bench_prof.json 2> $out/bench_prof.err
bash profiles/collect_pmc.sh $out/pmc > $out/pmc.log 2>&1
bash tools/pmc_fused.sh $out/pmc_fused > $out/pmc_fused.log 2>&1
python profiles/pmc_summary.py $out/pmc > $out/pmc_traffic.json
cp $(find $out/prof -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv
cat $out/bench.json
head -8 $out/kernel_stats.csv | cut -c1-160
cat $out/pmc_traffic.json
