#!/bin/bash
# usage (GPU box): tools/pmc_bench.sh <tag> [lib]  -- HBM fetch / write bytes and L2 hits of the bench loop's kernels (decode right behind an encode)
tag=$1; lib=$2
export TMPDIR=/tmp
[ -n "$lib" ] && export TRPX_LIB=$lib
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/${tag}_b$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/${tag}_b$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections
for i in (1,2,3):
    fs=glob.glob(f"gpurun_out/${tag}_b{i}/*/*counter_collection.csv")
    if not fs: print("pass", i, "failed"); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"].split("(")[0]
        if "k_decode_frames<" in k or "k_encode_fused" in k:
            acc[(k[:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k,c),v in sorted(acc.items()): print(f"${tag} {k:40s} {c:14s} n={len(v):3d} mean={sum(v)/len(v):14.1f} min={min(v):14.1f} max={max(v):14.1f}")
PY
