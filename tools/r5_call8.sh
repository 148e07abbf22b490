#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -x -k "fused_adam" > $O/t_halves.log 2>&1; echo "pytest rc=$?"; tail -2 $O/t_halves.log
B="python bench.py --no-roofline --no-cpu-baseline --sustain 0 --repeats 3 --steps 50 --warmup 10"
run() { echo "== $1"; shift; timeout -k 10 200 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; }
for rep in 1 2 3; do
run h0
run h64 --opt wgrad.adam_halves_max_rows=64
run h256 --opt wgrad.adam_halves_max_rows=256
run h1024 --opt wgrad.adam_halves_max_rows=1024
done > $O/ab8.txt 2>&1
cat $O/ab8.txt
for v in 0 64 256; do echo "== cyc b1 halves $v"; timeout -k 10 200 $B --model cyclegan --batch 1 --opt wgrad.adam_halves_max_rows=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; done
