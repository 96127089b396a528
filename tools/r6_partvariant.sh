#!/bin/bash
# round 6: a variant library with decode_part.hip compiled with extra flags.  usage: tools/r6_partvariant.sh <name> [-D...]
set -e
name=$1; shift
cd "$(dirname "$0")/../trpx_amd/csrc"
mkdir -p ../../tools/variants
/opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include "$@" -c decode_part.hip -o /tmp/trpx_pv_$name.o 2>&1 | grep -E "error" || true
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/variants/libtrpx_$name.so encode.o encode_fused.o decode.o decode_fast.o decode_frame.o decode_dense.o decode_seg.o shard.o bench_util.o api.o header_text.o /tmp/trpx_pv_$name.o -ldl
