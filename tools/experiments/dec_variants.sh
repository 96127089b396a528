#!/bin/bash
# usage (GPU box): tools/dec_variants.sh "<name>:<-D flags>" ...   builds decode_frame.hip variants, times tools/dec_time.py with each
# (three rounds, the variants alternating: run-to-run noise is 2-3 %)
cd "$(dirname "$0")/../trpx_amd/csrc"
mkdir -p ../../tools/variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -I../../include $flags -c decode_frame.hip -o /tmp/df_$name.o 2>/dev/null || { echo "$name: build failed"; continue; }
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/variants/libtrpx_$name.so encode.o encode_fused.o decode.o decode_fast.o /tmp/df_$name.o decode_seg.o shard.o api.o header_text.o -ldl
done
for rep in 1 2 3; do
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    echo -n "$name [$flags]: "; TRPX_LIB=$(pwd)/../../tools/variants/libtrpx_$name.so python3 ../../tools/dec_time.py 2>&1 | tail -1
  done
done
