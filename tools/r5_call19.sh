#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -q -x -k "column_owner or slab_reduce or finishes_the_layer or layer_stack" > $O/t_own.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/t_own.log
[ $rc -ne 0 ] && exit 1
B="python bench.py --no-roofline --no-cpu-baseline --sustain 0 --repeats 3 --steps 50 --warmup 10"
run() { echo "== $1"; shift; timeout -k 10 200 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; }
for rep in 1 2 3; do
run own16
run own0 --opt conv.own_max_rows=0
run own64kb512 --opt conv.own_max_rows=64 --opt conv.own_max_kb=512
done > $O/ab19.txt 2>&1
cat $O/ab19.txt
for v in 0 16; do echo "== cyc b1 own $v"; timeout -k 10 200 $B --model cyclegan --batch 1 --opt conv.own_max_rows=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; done
for v in 0 16; do echo "== cyc b4 own $v"; timeout -k 10 200 $B --model cyclegan --batch 4 --opt conv.own_max_rows=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; done
out=$O/own_prof; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/cls -o t -- python3 $R/tools/class_profile.py run --map $out/launch_map.json --steps 20 > $out/run.log 2>&1
python3 $R/tools/class_profile.py join $out/cls/t_kernel_trace.csv $out/launch_map.json $out/own16 > $out/join.log 2>&1
grep -E "conv_own" $out/own16_launches.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/cg -o cg -- python3 $R/bench.py --model cyclegan --batch 1 --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --sustain 0 --no-roofline > $out/cg.json 2> $out/cg.err
grep -E "conv_own" $out/cg/cg_kernel_stats.csv | cut -c1-200
rm -rf $out/cls $out/cg/*trace*
