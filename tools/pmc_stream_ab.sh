#!/bin/bash
# L2 traffic of k_bq_stream under library variants (tools/build_file_variant.sh ssmq_bq_stream NAME "-DBQS_..."): FETCH_SIZE and
# the TCC hit / miss counts, one rocprofv3 --pmc pass each, no tracing.  usage: tools/pmc_stream_ab.sh NAME ...
export TMPDIR=/tmp
for v in "$@"; do
  out=gpurun_out/pmc_ab_$v
  rm -rf $out; mkdir -p $out
  if [ "$v" = base ]; then unset SSMQ_LIBRARY; else export SSMQ_LIBRARY=variants/libssmq_$v.so; fi
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 tools/c5_deg7.py > $out/f.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/tcc -- python3 tools/c5_deg7.py > $out/t.log 2>&1
  grep "per transform" $out/f.log
  python3 - $out $v <<'PY'
import csv,glob,collections,sys
out,v=sys.argv[1],sys.argv[2]
for d in ("fetch","tcc"):
    acc=collections.defaultdict(list)
    for p in glob.glob(out+"/%s/**/*counter_collection.csv"%d, recursive=True):
        by=collections.defaultdict(float); nm={}
        for r in csv.DictReader(open(p)):
            by[(r["Dispatch_Id"],r["Counter_Name"])]+=float(r["Counter_Value"]); nm[r["Dispatch_Id"]]=r["Kernel_Name"][:40]
        for (did,c),val in by.items(): acc[(nm[did],c)].append(val)
    for k,val in sorted(acc.items()):
        if "bq_stream" in k[0]: print(v,d,k[1],len(val),"%.4g"%(sum(val)/len(val)))
PY
done
