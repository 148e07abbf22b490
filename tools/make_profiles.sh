#!/bin/bash
# Round profiles of the headline bench (run on the GPU box through gpurun; outputs under gpurun_out/profiles_<tag>/, the
# summaries are copied into profiles/ afterwards): kernel-trace statistics and separate PMC passes (FETCH_SIZE, WRITE_SIZE,
# SQ_*), exactly the command the driver runs with fewer steps.
tag=${1:-r05}
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --sustain 0 --no-roofline"
python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $out/bench_p16.json 2> $out/bench.err
# kernel statistics on the driver's own command line (20 timed steps); --no-roofline: the trace holds the 25 replays of the captured
# step (20 timed + 5 warm-up) + ONE eager warm-up step of capture(), no instrumented eager passes: per-step figures = totals / 26
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --sustain 0 --no-roofline > $out/bench_under_rocprof.json 2>> $out/bench.err
# the same with every launch on ONE stream (--single-stream): per-kernel durations without the other lanes' kernels sharing the chip
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1 -o s1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --sustain 0 --no-roofline --single-stream > $out/bench_single_stream_under_rocprof.json 2>> $out/bench.err
# per-CLASS table of the replayed single-stream step: every launch with its bench.py class label and algorithmic FLOPs beside the measured
# duration (tools/class_profile.py joins the library's launch log with the kernel trace, symbol by symbol)
rocprofv3 --kernel-trace --output-format csv -d $out/cls -o t -- python3 $GRAFT_REPO_ROOT/tools/class_profile.py run --map $out/launch_map.json --steps 20 > $out/class_profile_run.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/class_profile.py join $out/cls/t_kernel_trace.csv $out/launch_map.json $out/${tag}_bench_p16_single_stream > $out/class_profile_join.log 2>&1
cp $out/stats/s_kernel_stats.csv $out/${tag}_bench_p16_kernel_stats.csv
cp $out/stats1/s1_kernel_stats.csv $out/${tag}_bench_p16_kernel_stats_single_stream.csv
# CycleGAN 256x256 batch 1 (BASELINE.json configs[3] per-GPU shape): the two-chain schedule under the same profiler
rocprofv3 --kernel-trace --stats --output-format csv -d $out/statscg -o cg -- python3 $GRAFT_REPO_ROOT/bench.py --model cyclegan --batch 1 --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --sustain 0 --no-roofline > $out/bench_cyclegan_b1_under_rocprof.json 2>> $out/bench.err
cp $out/statscg/cg_kernel_stats.csv $out/${tag}_bench_cyclegan_b1_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- $CMD > /dev/null 2>> $out/bench.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- $CMD > /dev/null 2>> $out/bench.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d $out/sq -o q -- $CMD > /dev/null 2>> $out/bench.err
python3 $GRAFT_REPO_ROOT/tools/summarize_pmc.py $out $tag
ls -la $out
