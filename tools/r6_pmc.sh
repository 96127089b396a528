#!/bin/bash
# round 6: memory counters of one kernel of a tools/leg_prof.py leg (separate passes).  usage: tools/r6_pmc.sh <tag> <kernel substring> <leg> <mode> [TRPX_LIB]
tag=$1; kern=$2; leg=$3; mode=$4
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum"; do
  d=gpurun_out/$tag/$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $d -- python3 tools/leg_prof.py $leg $mode 3 > $d.log 2>&1
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/$tag/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "$kern" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
PY
