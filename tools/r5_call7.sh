#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
B="python bench.py --no-roofline --no-cpu-baseline --sustain 0 --repeats 3 --steps 50 --warmup 10"
run() { echo "== $1"; shift; timeout -k 10 200 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; }
for rep in 1 2 3; do
GAN_AMD_LIB=$R/gan_amd/libgan_amd_base.so run old_divs
run ns3_fastdiv
GAN_AMD_LIB=$R/gan_amd/libgan_amd_ns2.so run ns2
GAN_AMD_LIB=$R/gan_amd/libgan_amd_ns2.so run cyc1_ns2 --model cyclegan --batch 1
run cyc1_ns3 --model cyclegan --batch 1
done > $O/ab7.txt 2>&1
cat $O/ab7.txt
