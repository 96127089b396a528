#!/bin/bash
# usage (GPU box, repo root): tools/r5_ab.sh <lib-variant-name|default> <leg:mode> ...
v=$1; shift
for lm in "$@"; do
  leg=${lm%%:*}; mode=${lm##*:}
  if [ "$v" = default ]; then python3 tools/leg_prof.py $leg $mode 10 2>&1 | grep "ms per call" | sed "s/^/[default] /";
  else TRPX_LIB=tools/variants/libtrpx_$v.so python3 tools/leg_prof.py $leg $mode 10 2>&1 | grep "ms per call" | sed "s/^/[$v] /"; fi
done
