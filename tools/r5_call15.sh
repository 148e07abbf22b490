#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py tests/test_gpu_golden_full.py -q -x -k "statistics_partials or fused_backward or norm_fuse or pix2pix or benchmarked or p16 or p512 or cyclegan_train_step" > $O/t_rg.log 2>&1; echo "pytest rc=$?"; tail -3 $O/t_rg.log
B="python bench.py --no-roofline --no-cpu-baseline --sustain 0 --repeats 3 --steps 50 --warmup 10"
run() { echo "== $1"; shift; timeout -k 10 200 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; }
for rep in 1 2 3; do
run rg16
run rg1 --opt conv.reduce_stats_rg=1
run rg4 --opt conv.reduce_stats_rg=4
done > $O/ab15.txt 2>&1
cat $O/ab15.txt
for v in 16 1; do echo "== cyc b4 rg $v"; timeout -k 10 200 $B --model cyclegan --batch 4 --opt conv.reduce_stats_rg=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; done
for v in 16 1; do echo "== p512 b8 rg $v"; timeout -k 10 200 $B --img-size 512 --batch 8 --opt conv.reduce_stats_rg=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; done
