#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py tests/test_gpu_golden_full.py tests/test_gpu_configs.py -q -x -k "fused_backward or conv2d_fwd_dgrad or convT2d or pix2pix or benchmarked or p16 or c4 or cyclegan_train_step or fp16 or f16" > $O/t_thinbf.log 2>&1; echo "pytest rc=$?"; tail -3 $O/t_thinbf.log
B="python bench.py --no-roofline --no-cpu-baseline --sustain 0 --repeats 3 --steps 50 --warmup 10"
run() { echo "== $1"; shift; timeout -k 10 200 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; }
for rep in 1 2 3; do
run thinbf1
run thinbf0 --opt conv.thin_bwd_fuse=0
done > $O/ab16.txt 2>&1
cat $O/ab16.txt
for v in 1 0; do echo "== cyc b1 thinbf $v"; timeout -k 10 200 $B --model cyclegan --batch 1 --opt conv.thin_bwd_fuse=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; done
for v in 1 0; do echo "== cyc b16 thinbf $v"; timeout -k 10 200 $B --model cyclegan --batch 16 --steps 20 --opt conv.thin_bwd_fuse=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; done
