#!/bin/bash
# usage (GPU box, repo root): tools/experiments/idx_gap_tlb.sh <outfile>
# The indexed decoder's slow state (profiles/r04_idx_gap.txt) with the address-translation counters the round-4 passes lacked:
# idx_gap7.py under rocprofv3 --pmc, one process per counter set, the dispatches of k_decode_frames_indexed in script order,
# averaged per leg of 42 dispatches (each leg's first dropped), ratio to the first leg's in brackets.
out=$1
export TMPDIR=/tmp
tmp=gpurun_out/_tlb_$$
mkdir -p $tmp $(dirname $out)
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum SQ_WAVE_CYCLES SQ_WAIT_ANY" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --pmc $set --output-format csv -d $tmp/p$i -- python3 tools/experiments/idx_gap7.py > $tmp/p$i.log 2>&1 || echo "pass $i ($set) failed: $(tail -2 $tmp/p$i.log | tr '\n' ' ')"
  grep -h "with index" $tmp/p$i.log | cut -c1-110 | sed "s/^/pass $i: /" >> $tmp/legs.txt
done
python3 - $tmp > $out <<'PY'
import csv, glob, collections, sys
tmp = sys.argv[1]
print(open(f"{tmp}/legs.txt").read())
for f in sorted(glob.glob(f"{tmp}/p*/*/*counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "k_decode_frames_indexed" in r["Kernel_Name"]]
    by = collections.defaultdict(list)
    for r in rows: by[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for c, v in sorted(by.items()):
        v.sort()
        vals = [x for _, x in v]
        legs = [vals[i:i + 42] for i in range(0, len(vals), 42)]
        means = [sum(l[1:]) / max(1, len(l) - 1) for l in legs if len(l) > 1]
        print(f"{c:44s} n={len(vals):4d} " + "  ".join(f"{m:12.4g} ({m / means[0] if means[0] else 0:5.2f})" for m in means))
PY
rm -rf $tmp
cat $out
