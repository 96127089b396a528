#!/bin/bash
# usage (GPU box, repo root): tools/pmc_sets.sh <outfile> "<kernel name substring>" "<set1>;<set2>;..." <python script> [args]
# One rocprofv3 --pmc pass per ;-separated counter set (no trace domains next to --pmc; the program directly behind `--`); per pass:
# the script's own "ms per call" line and the means of the set's counters over the named kernel's launches.
out=$1; pat=$2; sets=$3; shift 3
export TMPDIR=/tmp
tmp=gpurun_out/_pmcs_$$
mkdir -p $tmp $(dirname $out)
: > $out
i=0
IFS=';' read -ra SETS <<< "$sets"
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout -k 10 90 rocprofv3 --pmc $set --output-format csv -d $tmp/p$i -- python3 "$@" > $tmp/p$i.log 2>&1 || echo "pass $i ($set) failed: $(tail -2 $tmp/p$i.log | tr '\n' ' ')" >> $out
  echo "pass $i: $(grep 'ms per call' $tmp/p$i.log | sed -E 's/ of 8 TB.*//')" >> $out
  python3 - $tmp/p$i "$pat" >> $out <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for f in sorted(glob.glob(f"{sys.argv[1]}/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    print(f"    {c:44s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
PY
done
rm -rf $tmp
cat $out
