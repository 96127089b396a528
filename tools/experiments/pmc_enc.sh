#!/bin/bash
# usage (GPU box): tools/pmc_enc.sh <tag> [lib]   -- SQ / LDS / TA counters of k_encode_fused (tools/enc_time.py)
tag=$1; lib=$2
export TMPDIR=/tmp
[ -n "$lib" ] && export TRPX_LIB=$lib
rm -rf gpurun_out/${tag}_p[0-9]*    # one run per directory (the summary below insists on it)
mkdir -p gpurun_out
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $set --output-format csv -d gpurun_out/${tag}_p$i -- python3 tools/enc_time.py > gpurun_out/${tag}_p$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections
for i in (1,2,3):
    fs=glob.glob(f"gpurun_out/${tag}_p{i}/*/*counter_collection.csv")
    if not fs: print("pass", i, "failed"); continue
    assert len(fs) == 1, f"gpurun_out/${tag}_p{i} holds {len(fs)} runs: clear it and run again"
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "k_encode_fused" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()): print(f"{k:36s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
PY
