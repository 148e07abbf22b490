#!/bin/bash
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r5/own_prof; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/cls -o t -- python3 $R/tools/class_profile.py run --map $out/launch_map.json --steps 20 > $out/run.log 2>&1
python3 $R/tools/class_profile.py join $out/cls/t_kernel_trace.csv $out/launch_map.json $out/own16 > $out/join.log 2>&1
grep -E "conv_own|16,128|64,128" $out/own16_launches.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/cg -o cg -- python3 $R/bench.py --model cyclegan --batch 1 --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --sustain 0 --no-roofline > $out/cg.json 2> $out/cg.err
grep -E "conv_own|splitk_norm|conv_gemm_kernel" $out/cg/cg_kernel_stats.csv | cut -c1-200
rm -rf $out/cls $out/cg/*trace*
