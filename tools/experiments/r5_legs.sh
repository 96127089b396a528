#!/bin/bash
# usage (GPU box, repo root): tools/r5_legs.sh <outdir> <leg:mode> ...   -- tools/leg_prof.py per leg, default route and round 4's parts route
out=$1; shift
mkdir -p $out
for lm in "$@"; do
  leg=${lm%%:*}; mode=${lm##*:}
  timeout -k 10 200 python3 tools/leg_prof.py $leg $mode 10 > $out/${leg}_${mode}.log 2>&1 || { echo "$lm failed"; tail -5 $out/${leg}_${mode}.log; exit 1; }
  echo "== new   $(grep 'ms per call' $out/${leg}_${mode}.log)"
  if [ "$mode" = free ]; then
    TRPX_DECODE_PATH=parts timeout -k 10 200 python3 tools/leg_prof.py $leg $mode 10 > $out/${leg}_${mode}_parts.log 2>&1 || { echo "$lm parts failed"; tail -5 $out/${leg}_${mode}_parts.log; exit 1; }
    echo "== parts $(grep 'ms per call' $out/${leg}_${mode}_parts.log)"
  fi
done
