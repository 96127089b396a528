#!/bin/bash
# usage (GPU box, repo root): tools/experiments/route_sweep.sh  -- per-frame route against the index route by frame size and count
for kind in p3 synth; do
  for geo in "640 640 1280" "768 768 888" "1024 1024 500" "1448 1448 250" "1030 1065 200" "1030 1065 480" "1030 1065 960" "2048 2048 128" "2048 2048 256"; do
    for rule in default "1,400000"; do
      if [ "$rule" = default ]; then unset TRPX_SINGLE_PART; else export TRPX_SINGLE_PART=$rule; fi
      timeout -k 10 120 python3 tools/experiments/route_sweep.py $kind $geo 2>&1 | grep -v amdgpu.ids | tail -1
    done
  done
done
