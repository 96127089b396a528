#!/bin/bash
# HBM traffic of the encode / decode kernels from the TCC counters (separate passes), with a calibration
# run of tools/membench (known byte counts, same 24-byte-per-lane access shape).  Run on the GPU box.
export TMPDIR=/tmp
out=gpurun_out/traffic
mkdir -p $out
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/cal_fetch -- ./tools/membench > $out/cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/cal_write -- ./tools/membench > $out/cal_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/enc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/enc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/enc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/enc_write.log 2>&1
python3 - <<PY | tee $out/summary.txt
import csv,glob,collections
def load(d):
    f=glob.glob(f"$out/{d}/*/*counter_collection.csv")[0]
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return acc
for d in ("cal_fetch","cal_write","enc_fetch","enc_write"):
    for (k,c),v in sorted(load(d).items()):
        if any(x in k for x in ("read16","read24","copy16","k_encode_fused","k_stitch","k_zero_words","k_walk_lds","k_unpack_tiles","k_decode_frames","k_synth")):
            print(f"{d:10s} {k:62s} {c:11s} n={len(v):3d} mean={sum(v)/len(v):16.1f} min={min(v):16.1f}")
PY
