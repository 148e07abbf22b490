#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
bash tools/ab.sh $O/ab33.txt "" "--step-attr dreal_on_side_lane=False" "--step-attr wgrad_alt=()" "--step-attr wgrad_cuts=(4,8)" "--opt conv.split_target_256=256" "--opt conv.big_min_blocks=32"
