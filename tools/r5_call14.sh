#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_step.py tests/test_gpu_golden_full.py -q -x -k "pix2pix or ddp or benchmarked or p16 or capture" > $O/t_head.log 2>&1; echo "pytest rc=$?"; tail -3 $O/t_head.log
B="python bench.py --no-roofline --no-cpu-baseline --sustain 0 --repeats 3 --steps 50 --warmup 10"
run() { echo "== $1"; shift; timeout -k 10 200 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; }
for rep in 1 2 3; do
run head_split
run head_one --step-attr head_on_side_lane=False
done > $O/ab14.txt 2>&1
cat $O/ab14.txt
