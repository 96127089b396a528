#!/bin/bash
# usage: tools/pmc_kernel.sh <tag> <kernel-substring> -- <python args...>   (run on the GPU box)
tag=$1; kern=$2; shift 3
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d gpurun_out/${tag}_p1 -- python3 "$@" > gpurun_out/${tag}_p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/${tag}_p2 -- python3 "$@" > gpurun_out/${tag}_p2.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in ("${tag}_p1","${tag}_p2"):
    f=glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv")[0]
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "${kern}" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()): print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):14.1f}")
PY
