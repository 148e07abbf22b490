#!/bin/bash
# final numbers: default bench line, the BASELINE configurations, the round's profile set
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout -k 10 200 python bench.py --gemm-only > $O/bench_default_final.json 2> $O/bench_default_final.err; echo "bench rc=$?"
timeout -k 10 500 bash tools/bench_configs.sh r05 > $O/bench_configs.log 2>&1; echo "configs rc=$?"; tail -9 $O/bench_configs.log
timeout -k 10 100 python tools/cpu_threads_sweep.py > $O/cpu_threads.txt 2>&1; cat $O/cpu_threads.txt
