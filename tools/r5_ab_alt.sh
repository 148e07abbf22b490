#!/bin/bash
# A/B of the second wgrad lane (Pix2PixStep.wgrad_alt), interleaved, one call
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
B="python bench.py --no-roofline --no-cpu-baseline --sustain 0 --repeats 3 --steps 50 --warmup 10"
run() { echo "== $1"; shift; timeout -k 10 200 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; }
for rep in 1 2; do
run base
run tail4 "--step-attr=wgrad_alt=('down3.kernel','down2.kernel','down1.kernel','down0.kernel')"
run up4 "--step-attr=wgrad_alt=('up0.kernel','up1.kernel','up2.kernel','up3.kernel')"
run both "--step-attr=wgrad_alt=('down3.kernel','down2.kernel','down1.kernel','down0.kernel','up0.kernel','up1.kernel','up2.kernel','up3.kernel')"
run adam_d "--step-attr=wgrad_alt=('down4.kernel','down5.kernel','down6.kernel','down7.kernel')"
run pp "--step-attr=wgrad_alt=('up4.kernel','up5.kernel','up6.kernel')"
done > $O/ab_alt.txt 2>&1
cat $O/ab_alt.txt
timeout -k 10 1000 python -m pytest tests/test_gpu_golden_full.py tests/test_gpu_ops.py -k "golden or oracle_values or never_read" -q -s > $O/t_new2.log 2>&1; echo "pytest rc=$?" >> $O/t_new2.log
tail -30 $O/t_new2.log | cut -c1-300
