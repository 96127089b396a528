#!/bin/bash
# Round profile set (run on the GPU box from the repo root): bench line, rocprofv3 kernel stats of the same command, PMC
# traffic passes (FETCH_SIZE / WRITE_SIZE separately, with the tools/membench calibration), all under gpurun_out/$1.
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
rm -rf $out/stats $out/cal_fetch $out/cal_write $out/enc_fetch $out/enc_write   # one run per directory: the globs below must have exactly one match
export TMPDIR=/tmp
python3 bench.py > $out/bench.json 2> $out/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --headline-only > $out/bench_under_rocprof.json 2> $out/rocprof.err || exit 1
[ $(ls $out/stats/*/*kernel_stats.csv | wc -l) -eq 1 ] || { echo "profile_round: $out/stats holds more than one run" >&2; exit 1; }
cp $out/stats/*/*kernel_stats.csv $out/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/cal_fetch -- ./tools/membench > $out/cal_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/cal_write -- ./tools/membench > $out/cal_write.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/enc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --headline-only > $out/enc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/enc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --headline-only > $out/enc_write.log 2>&1 || exit 1
python3 - <<PY | tee $out/pmc_traffic.txt
import csv,glob,collections
def load(d):
    fs=glob.glob(f"$out/{d}/*/*counter_collection.csv")
    assert len(fs) == 1, f"$out/{d} holds {len(fs)} runs: clear it and run again"
    f=fs[0]
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return acc
for d in ("cal_fetch","cal_write","enc_fetch","enc_write"):
    for (k,c),v in sorted(load(d).items()):
        if any(x in k for x in ("read16","read24","copy16","k_encode_fused","k_stitch","k_zero_words","k_seg","k_unpack","k_decode_frames","k_synth","k_walk")):
            print(f"{d:10s} {k:62s} {c:11s} n={len(v):3d} mean={sum(v)/len(v):16.1f} min={min(v):16.1f} max={max(v):16.1f}")
PY
