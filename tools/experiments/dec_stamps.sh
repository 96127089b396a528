#!/bin/bash
# usage (GPU box): tools/dec_stamps.sh "<-D flags>"   -- TRPX_DEC_STAMPS build of decode_frame.hip + tools/dec_stamps.py
cd "$(dirname "$0")/../trpx_amd/csrc"
mkdir -p ../../tools/variants
/opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -I../../include -DTRPX_DEC_STAMPS $1 -c decode_frame.hip -o /tmp/df_st.o 2>/dev/null || exit 1
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/variants/libtrpx_stamps.so encode.o encode_fused.o decode.o decode_fast.o /tmp/df_st.o decode_seg.o shard.o api.o header_text.o -ldl
cd ../..
echo "== stamps [$1]"
TRPX_LIB=$PWD/tools/variants/libtrpx_stamps.so python3 tools/dec_stamps.py 2>&1 | grep -v amdgpu.ids
