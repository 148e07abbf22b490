#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/full_gpu_tests2.log 2>&1; echo "pytest rc=$?"; tail -4 $O/full_gpu_tests2.log
B="python bench.py --no-roofline --no-cpu-baseline --sustain 0 --repeats 3 --steps 50 --warmup 10"
run() { echo "== $1"; shift; timeout -k 10 200 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; }
for rep in 1 2; do
run base
run ppgf15 --opt wgrad.pingpong_min_gflop=15
run cuts_4_8_11 "--step-attr=wgrad_cuts=(4,8,11)"
run alt_d2 "--step-attr=wgrad_alt=('down2.kernel','down1.kernel','down0.kernel')"
run alt_d4 "--step-attr=wgrad_alt=('down4.kernel','down3.kernel','down2.kernel','down1.kernel','down0.kernel')"
done > $O/ab5.txt 2>&1
cat $O/ab5.txt
