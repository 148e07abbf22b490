#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "norm_finalize_inside or norm_act_fwd_bwd or fused_backward" > $O/t28_ops.txt 2>&1; echo "ops rc=$?"; tail -5 $O/t28_ops.txt
timeout -k 10 900 python -m pytest tests/test_gpu_step.py -x -q -m gpu > $O/t28_step.txt 2>&1; echo "step rc=$?"; tail -5 $O/t28_step.txt
bash tools/ab.sh $O/ab28.txt "--opt norm.fin_in_apply=0" "" 
