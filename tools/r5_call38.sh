#!/bin/bash
# one-rank RCCL rehearsal of the data-parallel schedule: schedule constants on the final tree
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
export GAN_AMD_DDP_REHEARSE=1
bash tools/ab.sh $O/ab38.txt "" "--step-attr ddp_graphs=3" "--step-attr ddp_graphs=2" "--step-attr ddp_late_comm=False" "--step-attr ddp_buckets=False" "--step-attr dreal_on_side_lane=False" "--fp32-allreduce"
