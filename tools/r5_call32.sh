#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/t32_all.txt 2>&1; echo "all rc=$?"; tail -5 $O/t32_all.txt
