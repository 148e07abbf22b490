#!/bin/bash
# round-5 first GPU call: new tests, default bench (shipped-plan roofline), class profile, multi-lane trace + eager lane timeline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_golden_full.py tests/test_gpu_ops.py -k "golden or oracle_values or never_read" -x -q -s > $O/t_new.log 2>&1; echo "pytest rc=$?" >> $O/t_new.log
tail -5 $O/t_new.log
timeout -k 10 300 python bench.py --gemm-only > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/cls -o t -- python3 $R/tools/class_profile.py run --map $O/launch_map.json --steps 20 > $O/cls_run.log 2>&1; echo "cls rc=$?"
python3 $R/tools/class_profile.py join $O/cls/t_kernel_trace.csv $O/launch_map.json $O/r05_bench_p16_single_stream > $O/cls_join.log 2>&1; echo "join rc=$?"
tail -20 $O/cls_join.log
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o tr -- python3 $R/bench.py --no-cpu-baseline --no-roofline --sustain 0 --repeats 1 --steps 6 --warmup 3 > $O/trace_bench.json 2> $O/trace_bench.err; echo "trace rc=$?"
cd $R
timeout -k 10 200 python tools/lane_timeline.py > $O/lane_timeline.txt 2>&1; echo "timeline rc=$?"
ls $O
