#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "statistics_partials or fused_backward or never_read_past" > $O/t35_ops.txt 2>&1; echo "ops rc=$?"; tail -3 $O/t35_ops.txt
timeout -k 10 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_configs.py -x -q -m gpu > $O/t35_step.txt 2>&1; echo "step rc=$?"; tail -3 $O/t35_step.txt
