#!/bin/bash
# usage (GPU box): tools/ab_rt.sh <variant> ...   -- round trip (tools/rt_idx.py, plain legs) with tools/variants/libtrpx_<variant>.so, twice, alternating; "product" = the in-tree library
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = product ]; then lib=""; else lib=$PWD/tools/variants/libtrpx_$v.so; fi
  echo "$v: $(TRPX_LIB=$lib timeout -k 10 200 python3 tools/rt_idx.py 2>&1 | grep plain | sed -e 's/plain //' -e 's/ exact True//' | tr '\n' '|')"
done; done
