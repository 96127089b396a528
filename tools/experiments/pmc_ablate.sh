#!/bin/bash
export TMPDIR=/tmp
for l in trpx_amd/libtrpx_hip.so tools/variants/libtrpx_ab1.so tools/variants/libtrpx_ab4.so tools/variants/libtrpx_ab5.so; do
  tag=$(basename $l .so)
  TRPX_LIB=$l rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pv_$tag -- python3 tools/enc_time.py > gpurun_out/pv_$tag.log 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pv_$tag/*/*counter_collection.csv")[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_encode_fused" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$tag", {k: round(sum(v)/len(v)/1e6,1) for k,v in sorted(acc.items())})
PY
done
