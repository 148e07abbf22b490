#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
timeout -k 10 600 python tools/soak_determinism.py > $O/soak.txt 2>&1; echo "soak rc=$?"; tail -6 $O/soak.txt
