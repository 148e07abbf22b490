#!/bin/bash
# usage: tools/ablate.sh "<bench_op args>"  - times one conv GEMM with parts of the kernel disabled (GAN_AMD_GEMM_DEBUG bits:
# 1 no MFMA phase, 2 no loads, 4 no epilogue, 8 no gather table, 16/32 no epilogue staging / stores (n/a), 64 no per-step
# barrier (only meaningful together with 2))
for dbg in 0 1 2 3 4 5 6 7 12 15; do
  echo -n "debug=$dbg  "
  GAN_AMD_GEMM_DEBUG=$dbg python3 tools/bench_op.py $1 50 2>/dev/null | grep -v amdgpu.ids
done
