#!/bin/bash
# usage (GPU box): tools/kstats.sh <tag> <python script> [args]  -- per-kernel durations (rocprofv3 kernel trace) of a tools/ script
tag=$1; shift
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -- python3 "$@" > gpurun_out/$tag.log 2>&1
f=$(ls gpurun_out/$tag/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cut -d, -f1-4 "$f" | cut -c1-60,100-200 | head -16
