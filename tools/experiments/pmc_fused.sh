#!/bin/bash
# usage: tools/pmc_fused.sh <tag> [frames]   (run on the GPU box; writes gpurun_out/<tag>_pmc.txt)
tag=$1; frames=${2:-200}
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d gpurun_out/${tag}_p1 -- python3 tools/enc_time.py $frames > gpurun_out/${tag}_p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/${tag}_p2 -- python3 tools/enc_time.py $frames > gpurun_out/${tag}_p2.log 2>&1
python3 - <<PY > gpurun_out/${tag}_pmc.txt
import csv,glob,collections
for d in ("${tag}_p1","${tag}_p2"):
    f=glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv")[0]
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_encode_fused" in r["Kernel_Name"] or "k_pack" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()): print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):14.1f}")
PY
cat gpurun_out/${tag}_pmc.txt
