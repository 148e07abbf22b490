#!/bin/bash
# kernel trace of the one-rank RCCL rehearsal of the data-parallel schedule -> gpurun_out/prof_ddp/
cd /tmp && export TMPDIR=/tmp
export GAN_AMD_DDP_REHEARSE=1
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ddp -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --repeats 1 --no-cpu-baseline > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/prof_ddp | head
