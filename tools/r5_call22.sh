#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
B="python bench.py --no-roofline --no-cpu-baseline --sustain 0 --repeats 3 --steps 50 --warmup 10"
run() { echo "== $1"; shift; timeout -k 10 200 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; }
for rep in 1 2; do
run default
run minrows2048 --opt wgrad.pingpong_min_rows=2048
run minrows4096 --opt wgrad.pingpong_min_rows=4096
run minrows8192 --opt wgrad.pingpong_min_rows=8192
done > $O/ab22.txt 2>&1
cat $O/ab22.txt
