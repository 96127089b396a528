#!/bin/bash
# usage (GPU box, repo root): tools/ab_legs.sh "<variants: product or names of tools/variants/libtrpx_<name>.so>" <leg:mode> ...
vars=$1; shift
for lm in "$@"; do
  leg=${lm%%:*}; mode=${lm##*:}
  for v in $vars; do
    if [ "$v" = product ]; then lib=""; else lib=$PWD/tools/variants/libtrpx_$v.so; fi
    echo "$v: $(TRPX_LIB=$lib timeout -k 10 200 python3 tools/leg_prof.py $leg $mode 10 2>&1 | grep 'ms per call')"
  done
done
