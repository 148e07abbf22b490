#!/bin/bash
# A/B of the small-tile pipeline depth (libgan_amd_base.so = 3 / 2 stages, libgan_amd.so = 5 stages), op tests on the new library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -q -x -k "conv or stack or split or norm" > $O/t_ns5.log 2>&1; echo "pytest rc=$?"; tail -3 $O/t_ns5.log
B="python bench.py --no-roofline --no-cpu-baseline --sustain 0 --repeats 3 --steps 50 --warmup 10"
run() { echo "== $1"; shift; timeout -k 10 200 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; }
for rep in 1 2 3; do
GAN_AMD_LIB=$R/gan_amd/libgan_amd_base.so run base
run ns5
GAN_AMD_LIB=$R/gan_amd/libgan_amd_base.so run cyc1_base --model cyclegan --batch 1
run cyc1_ns5 --model cyclegan --batch 1
done > $O/ab6.txt 2>&1
cat $O/ab6.txt
cd /tmp && export TMPDIR=/tmp
for v in base ns5; do
  if [ $v = base ]; then export GAN_AMD_LIB=$R/gan_amd/libgan_amd_base.so; else unset GAN_AMD_LIB; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/cls_$v -o t -- python3 $R/tools/class_profile.py run --map $O/launch_map_$v.json --steps 20 > $O/cls_run_$v.log 2>&1
  python3 $R/tools/class_profile.py join $O/cls_$v/t_kernel_trace.csv $O/launch_map_$v.json $O/cls_$v > $O/cls_join_$v.log 2>&1
  grep "conv_gemm<bf16,64\|conv_gemm<bf16,16\|sum of kernel" $O/cls_join_$v.log
done
