#!/bin/bash
# HBM traffic + SQ counters of the N = 1181 route (tools/c5_deg7.py: D = E = 10, B = 1e4; k_eval_wave + k_bq_stream): separate
# rocprofv3 --pmc passes, no tracing
export TMPDIR=/tmp
out=gpurun_out/pmc_s
rm -rf $out; mkdir -p $out
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 tools/c5_deg7.py > $out/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 tools/c5_deg7.py > $out/w.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/sq -- python3 tools/c5_deg7.py > $out/s.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $out/sq2 -- python3 tools/c5_deg7.py > $out/s2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_WAVE32_LDS --output-format csv -d $out/sq3 -- python3 tools/c5_deg7.py > $out/s3.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/tcc -- python3 tools/c5_deg7.py > $out/t.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in ("fetch","write","sq","sq2","sq3","tcc"):
    acc=collections.defaultdict(list)
    for p in glob.glob("$out/%s/**/*counter_collection.csv"%d, recursive=True):
        by=collections.defaultdict(float); nm={}
        for r in csv.DictReader(open(p)):
            by[(r["Dispatch_Id"],r["Counter_Name"])]+=float(r["Counter_Value"]); nm[r["Dispatch_Id"]]=r["Kernel_Name"][:48]
        for (did,c),v in by.items(): acc[(nm[did],c)].append(v)
    for k,v in sorted(acc.items()):
        if "bq_stream" in k[0] or "aos_to" in k[0] or "eval_wave" in k[0]: print(d,k,len(v),sum(v)/len(v))
PY
