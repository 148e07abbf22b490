#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_golden_full.py -q -x -k "pix2pix or benchmarked or p16 or c4 or cyclegan_train_step" > $O/t_own2.log 2>&1; rc=$?; echo "pytest2 rc=$rc"; tail -5 $O/t_own2.log
[ $rc -ne 0 ] && exit 1
B="python bench.py --no-roofline --no-cpu-baseline --sustain 0 --repeats 3 --steps 50 --warmup 10"
run() { echo "== $1"; shift; timeout -k 10 200 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; }
for rep in 1 2 3; do
run own16
run own0 --opt conv.own_max_rows=0
run own64 --opt conv.own_max_rows=64
done > $O/ab17.txt 2>&1
cat $O/ab17.txt
for v in 0 16 64; do echo "== cyc b1 own $v"; timeout -k 10 200 $B --model cyclegan --batch 1 --opt conv.own_max_rows=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; done
for v in 0 16 64; do echo "== cyc b4 own $v"; timeout -k 10 200 $B --model cyclegan --batch 4 --opt conv.own_max_rows=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; done
