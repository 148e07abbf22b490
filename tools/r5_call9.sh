#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q > $O/full_gpu_tests3.log 2>&1; echo "pytest rc=$?"; tail -3 $O/full_gpu_tests3.log
B="python bench.py --no-roofline --no-cpu-baseline --sustain 0 --repeats 3 --steps 50 --warmup 10"
run() { echo "== $1"; shift; timeout -k 10 200 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; }
for rep in 1 2; do
run dconc0
run dconc1 --step-attr d_wgrad_concurrent=1
run dconc2 --step-attr d_wgrad_concurrent=2
done > $O/ab9.txt 2>&1
cat $O/ab9.txt
