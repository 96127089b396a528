#!/bin/bash
# usage (GPU box, repo root): tools/leg_stats.sh <outdir> <leg:mode> [<leg:mode> ...]
# rocprofv3 --kernel-trace --stats of tools/leg_prof.py per leg; prints / keeps the top of each kernel_stats.csv
out=$1; shift
mkdir -p $out
export TMPDIR=/tmp
for lm in "$@"; do
  leg=${lm%%:*}; mode=${lm##*:}
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${leg}_${mode} -- python3 tools/leg_prof.py $leg $mode 10 > $out/${leg}_${mode}.log 2>&1 || { echo "$lm failed"; tail -3 $out/${leg}_${mode}.log; exit 1; }
  f=$(ls $out/${leg}_${mode}/*/*kernel_stats.csv | head -1)
  cp $f $out/${leg}_${mode}_kernel_stats.csv
  rm -rf $out/${leg}_${mode}
  echo "== $lm: $(grep 'ms per call' $out/${leg}_${mode}.log)"
  python3 - $out/${leg}_${mode}_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:7]:
    print(f"   {r['Name'].split('(')[0][:70]:70s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.2f} min={float(r['MinNs'])/1e3:9.2f} max={float(r['MaxNs'])/1e3:9.2f} pct={r['Percentage']}")
PY
done
