#!/bin/bash
# usage (GPU box, repo root): tools/ab_idx.sh  -- the indexed per-frame decode with / without the filler's stream touch
# (make -C trpx_amd/csrc idxtouch): the slow-state sequence of tools/experiments/idx_gap7.py, then the bench's legs.
for v in product notouch product notouch; do
  if [ "$v" = product ]; then lib=""; else lib=$PWD/tools/variants/libtrpx_$v.so; fi
  echo "== $v: idx_gap7"
  TRPX_LIB=$lib timeout -k 10 300 python3 tools/experiments/idx_gap7.py 2>&1 | grep -E "with index" | sed -E 's/\| new segments.*//'
done
for v in product notouch; do
  if [ "$v" = product ]; then lib=""; else lib=$PWD/tools/variants/libtrpx_$v.so; fi
  echo "== $v: bench legs"
  TRPX_LIB=$lib timeout -k 10 400 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline', round(b['value']), 'decode_with_index_ms', round(b['decode_with_index_ms'],4))
for k in ['noisy_u16','poisson3_u16','midsize_u16','midsize_poisson3_u16','oddsize_u16','config3_4096x4096_int32']:
    v=b[k]; print(k, 'dec', round(v['decode_ms'],4), 'idx', round(v['decode_with_index_ms'],4))
"
done
