#!/bin/bash
# A/B of single ops between this tree and a copy of an older commit unpacked + built under _old/ (same gpurun call)
for shp in "conv_fwd 16 128 64 128 2" "conv_fwd 16 64 128 256 2" "convT_fwd 16 32 512 128 2" "conv_fwd 32 64 128 256 2" "conv_fwd 16 33 512 256 1" "conv_fwd 32 32 256 512 1"; do
  echo "new: $(python3 tools/bench_op.py $shp 50 2>/dev/null | grep -v amdgpu.ids)"
  echo "old: $(cd _old && python3 tools/bench_op.py $shp 50 2>/dev/null | grep -v amdgpu.ids)"
done
