#!/bin/bash
# Produces the artefacts committed under profiles/: bench JSON, rocprofv3 kernel stats of the same command, PMC traffic.
set -e
export TMPDIR=/tmp
tag=${1:-r06_final}
out=gpurun_out/$tag
mkdir -p $out
# the tree these numbers belong to: tools/stamp_tree.sh writes TREE_ID before the gpurun call (there is no .git on the GPU box)
export SSMQ_COMMIT=$(cat TREE_ID 2>/dev/null || echo unknown)
python bench.py --detail $out/bench_detail.json > $out/bench.json 2> $out/bench.err      # bench.json: the result line (what the driver keeps)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --no-cpu-baseline --detail $out/bench_prof_detail.json > $out/bench_prof.json 2> $out/bench_prof.err
bash profiles/collect_pmc.sh $out/pmc > $out/pmc.log 2>&1
bash tools/pmc_fused.sh $out/pmc_fused > $out/pmc_fused.log 2>&1
python profiles/pmc_summary.py $out/pmc > $out/pmc_traffic.json
cp $(find $out/prof -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv
# the same trace split by grid size: bench.py launches some kernels at two batch sizes (the D = 6 transform at 1e5 and 1e6)
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for path in glob.glob(out + '/prof/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        g = r.get('Grid_Size_X') or r.get('Grid_Size')
        acc[(r['Kernel_Name'], int(g))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
with open(out + '/kernel_stats_by_grid.csv', 'w') as f:
    f.write('"Name","GridSize","Calls","AverageNs","MinNs","MaxNs"\n')
    for (name, grid), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        f.write('"%s",%d,%d,%.1f,%d,%d\n' % (name, grid, len(v), sum(v) / len(v), min(v), max(v)))
PY
tail -1 $out/bench.json
head -8 $out/kernel_stats.csv | cut -c1-160
cat $out/pmc_traffic.json
# performance gate against the newest committed trace (the previous round's or this round's earlier one), as a markdown table
prev=$(ls profiles/r0*_bench_kernel_stats_by_grid.csv 2>/dev/null | sort | tail -1)
echo "perf gate baseline: $prev"
if [ -n "$prev" ]; then
    python3 tools/perf_gate.py "$prev" $out/kernel_stats_by_grid.csv --markdown --matched-only > $out/perf_gate.md 2> $out/perf_gate.err || echo "perf_gate: REGRESSION (see $out/perf_gate.err)"
    cat $out/perf_gate.err
fi
