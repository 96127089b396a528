#!/bin/bash
# usage (GPU box, repo root): tools/pmc_kernels.sh <outfile> "<kernel name substrings, |-separated>" <python script> [args]
# SQ / LDS / TA / TCP counters of the named kernels, in separate rocprofv3 --pmc passes (no trace domains next to --pmc; the
# program directly behind `--`), means per kernel and counter.
out=$1; pat=$2; shift 2
export TMPDIR=/tmp
tmp=gpurun_out/_pmc_$$
mkdir -p $tmp $(dirname $out)
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $tmp/p$i -- python3 "$@" > $tmp/p$i.log 2>&1 || echo "pass $i ($set) failed: $(tail -2 $tmp/p$i.log | tr '\n' ' ')"
done
python3 - $tmp "$pat" > $out <<'PY'
import csv, glob, collections, sys
tmp, pat = sys.argv[1], sys.argv[2].split("|")
acc = collections.defaultdict(list)
for f in sorted(glob.glob(f"{tmp}/p*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for p in pat:
            if p in k:
                acc[(p, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (p, c), v in sorted(acc.items()):
    print(f"{p:32s} {c:36s} n={len(v):3d} mean={sum(v)/len(v):18.1f}")
PY
rm -rf $tmp
cat $out
