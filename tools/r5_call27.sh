#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
MARGINAL_VERBOSE=1 timeout -k 10 500 python tools/marginal.py > $O/marginal.txt 2> $O/marginal.err
grep "ms/step" $O/marginal.txt; tail -3 $O/marginal.err
