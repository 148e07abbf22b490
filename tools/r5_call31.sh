#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_step.py -x -q -m gpu > $O/t31_step.txt 2>&1; echo "step rc=$?"; tail -5 $O/t31_step.txt
bash tools/ab.sh $O/ab31.txt "--step-attr bias_grad_on_side=False --step-attr adam_begin_early=False" "" "--step-attr adam_begin_early=False" "--step-attr bias_grad_on_side=False"
