#!/bin/bash
# usage (GPU box): tools/enc_variants.sh "<name>:<-D flags>" ...   encode_fused.hip variants timed with tools/enc_time.py (3 rounds)
cd "$(dirname "$0")/../trpx_amd/csrc"
mkdir -p ../../tools/variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -I../../include $flags -c encode_fused.hip -o /tmp/ef_$name.o 2>/dev/null || { echo "$name: build failed"; continue; }
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/variants/libtrpx_$name.so encode.o /tmp/ef_$name.o decode.o decode_fast.o decode_frame.o decode_seg.o shard.o api.o header_text.o -ldl
done
cd ../..
for rep in 1 2 3; do
  for spec in "$@"; do
    name=${spec%%:*}
    echo -n "$name: "; TRPX_LIB=$PWD/tools/variants/libtrpx_$name.so python3 tools/enc_time.py 2>&1 | tail -1 | cut -c1-60
  done
done
