#!/bin/bash
# round 6: tools/fuzz_paths.py along every decode route (GPU box, repo root); $1 = output file
out=${1:-gpurun_out/r6_fuzz.txt}
: > $out
run() { TRPX_DECODE_PATH=$1 timeout -k 10 900 python3 tools/fuzz_paths.py $2 $3 2>&1 | grep -E "^OK|Error|assert|FAILED" >> $out || echo "path $1 FAILED" >> $out; }
run "" 200 6060
run dense 200 6061
run frames 100 6060
run tiles 100 6060
run parts 100 6060
run basic 100 6060
cat $out
