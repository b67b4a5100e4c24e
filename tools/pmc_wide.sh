#!/bin/bash
# SQ-level counters of the generic kernel's two passes in the D = 10, N = 201 transform (separate PMC passes)
out=gpurun_out/pmc_wide
export TMPDIR=/tmp
rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $out/a -- python3 tools/c5_full.py 10000 > $out/a.log 2>&1 || tail -3 $out/a.log
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/b -- python3 tools/c5_full.py 10000 > $out/b.log 2>&1 || tail -3 $out/b.log
python3 - <<'PY'
import csv, glob, collections
for sub in ('a', 'b'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob('gpurun_out/pmc_wide/' + sub + '/**/*counter_collection.csv', recursive=True):
        rows = list(csv.DictReader(open(path)))
        # the EVAL and FX passes alternate: tell them apart by dispatch order (EVAL first of each pair)
        order = {}
        for r in rows:
            if 'k_apply_wide' in r['Kernel_Name'] and int(r['Grid_Size']) >= 256 * 5000:
                did = int(r['Dispatch_Id'])
                order.setdefault(did, len(order))
        for r in rows:
            did = int(r['Dispatch_Id'])
            if did in order:
                acc['EVAL' if order[did] % 2 == 0 else 'FX'][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        print(sub, k)
        for c, v in sorted(d.items()):
            print('   %-24s %16.1f  (n=%d)' % (c, sum(v) / len(v), len(v)))
PY
