#!/bin/bash
# usage (GPU box): tools/enc_variant.sh <name> "<-D flags>"   -- variant of encode_fused.hip -> tools/variants/libtrpx_<name>.so
cd "$(dirname "$0")/../trpx_amd/csrc"
mkdir -p ../../tools/variants
/opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -I../../include $2 -c encode_fused.hip -o /tmp/ef_$1.o 2>/dev/null || { echo "$1: build failed"; exit 1; }
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/variants/libtrpx_$1.so encode.o /tmp/ef_$1.o decode.o decode_fast.o decode_frame.o decode_seg.o shard.o api.o header_text.o -ldl
