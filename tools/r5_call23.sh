#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -q -x > $O/t_ops.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/t_ops.log
[ $rc -ne 0 ] && exit 1
B="python bench.py --no-roofline --no-cpu-baseline --sustain 0 --repeats 3 --steps 50 --warmup 10"
run() { echo "== $1"; shift; timeout -k 10 200 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; }
run p16; run p16
echo "== cyc b1"; timeout -k 10 200 $B --model cyclegan --batch 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"
out=$O/own_prof; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/cls -o t -- python3 $R/tools/class_profile.py run --map $out/launch_map.json --steps 20 > $out/run.log 2>&1
python3 $R/tools/class_profile.py join $out/cls/t_kernel_trace.csv $out/launch_map.json $out/rg4 > $out/join.log 2>&1
tail -3 $out/join.log
rm -rf $out/cls
