#!/bin/bash
# usage (GPU box, repo root): tools/r5_kstats.sh <outdir> <leg:mode> ...  -- rocprofv3 kernel stats per leg (name, calls, total, avg)
out=$1; shift
mkdir -p $out
export TMPDIR=/tmp
for lm in "$@"; do
  leg=${lm%%:*}; mode=${lm##*:}
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_${leg}_${mode} -- python3 tools/leg_prof.py $leg $mode 10 > $out/${leg}_${mode}.log 2>&1 || { echo "$lm stats failed"; tail -3 $out/${leg}_${mode}.log; exit 1; }
  cp $(ls $out/t_${leg}_${mode}/*/*kernel_stats.csv | head -1) $out/${leg}_${mode}_kernel_stats.csv; rm -rf $out/t_${leg}_${mode}
  echo "== $lm: $(grep 'ms per call' $out/${leg}_${mode}.log)"
  python3 - $out/${leg}_${mode}_kernel_stats.csv <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("void trpx::", "")[:48]
    if r["Name"].startswith("void at::") or "elementwise" in r["Name"] or "synth" in r["Name"]: continue
    print(f"   {n:48s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}  max {float(r['MaxNs'])/1e3:8.1f}")
P
done
