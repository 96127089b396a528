#!/bin/bash
# usage (GPU box): tools/bench_variants.sh "<name>:<-D flags>" ...   decode_frame.hip variants, decode_frames ms inside bench.py's round trip
cd "$(dirname "$0")/../trpx_amd/csrc"
mkdir -p ../../tools/variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -I../../include $flags -c decode_frame.hip -o /tmp/df_$name.o 2>/dev/null || { echo "$name: build failed"; continue; }
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/variants/libtrpx_$name.so encode.o encode_fused.o decode.o decode_fast.o /tmp/df_$name.o decode_seg.o shard.o api.o header_text.o -ldl
done
cd ../..
for rep in 1 2 3; do
  for spec in "$@"; do
    name=${spec%%:*}
    TRPX_LIB=$PWD/tools/variants/libtrpx_$name.so python3 bench.py --headline-only --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value']), 'dec', round(d['kernel_ms']['decode_frames'],4), 'enc', round(d['kernel_ms']['encode_fused'],4), 'dec-only Mfps', round(d['decode_fps']/1e6,2))"
  done
done
