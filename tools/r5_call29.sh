#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
bash tools/prof_ab.sh gpurun_out/r5b/prof29 "--opt norm.fin_in_apply=0" ""
python tools/cmp_stats.py gpurun_out/r5b/prof29/v0_kernel_stats.csv gpurun_out/r5b/prof29/v1_kernel_stats.csv 26 > gpurun_out/r5b/prof29/cmp.txt
rm -f gpurun_out/r5b/prof29/*_kernel_trace.csv
cat gpurun_out/r5b/prof29/cmp.txt | cut -c1-200
