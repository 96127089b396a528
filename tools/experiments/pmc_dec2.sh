#!/bin/bash
# usage (GPU box): tools/pmc_dec2.sh <tag> <frames> [lib]  -- L2 hit/miss, fetch/write size, issue counters of k_decode_frames (tools/dec_time.py <frames>)
tag=$1; frames=$2; lib=$3
export TMPDIR=/tmp
[ -n "$lib" ] && export TRPX_LIB=$lib
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $set --output-format csv -d gpurun_out/${tag}_p$i -- python3 tools/dec_time.py $frames > gpurun_out/${tag}_p$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections
for i in (1,2,3,4):
    fs=glob.glob(f"gpurun_out/${tag}_p{i}/*/*counter_collection.csv")
    if not fs: print("pass", i, "failed"); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "k_decode_frames<" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()): print(f"${tag} {k:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
PY
