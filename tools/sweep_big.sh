#!/bin/bash
# usage: tools/sweep_big.sh  (env knobs are read per process)
for shp in "conv_fwd 16 33 512 256 1" "convT_fwd 16 16 1024 256 2" "conv_fwd 16 32 256 1024 2" "conv_fwd 16 64 128 256 2" "convT_fwd 16 16 512 256 2" "conv_fwd 16 128 64 128 2" "conv_fwd 32 128 64 128 2"; do
  python3 tools/bench_op.py $shp 50 2>/dev/null | grep -v amdgpu.ids
done
