#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
GAN_AMD_LIB=$R/gan_amd/libgan_amd_thinlds.so timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -x -k "conv2d_fwd_dgrad_wgrad or convT2d_fwd" > $O/t_thin.log 2>&1; echo "pytest rc=$?"; tail -2 $O/t_thin.log
B="python bench.py --no-roofline --no-cpu-baseline --sustain 0 --repeats 3 --steps 50 --warmup 10"
run() { echo "== $1"; shift; timeout -k 10 200 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; }
for rep in 1 2 3; do
run base
GAN_AMD_LIB=$R/gan_amd/libgan_amd_thinlds.so run thinlds
done > $O/ab13.txt 2>&1
cat $O/ab13.txt
for v in base thinlds; do
  if [ $v = base ]; then unset GAN_AMD_LIB; else export GAN_AMD_LIB=$R/gan_amd/libgan_amd_thinlds.so; fi
  python tools/profile_ops.py 2>/dev/null | grep "conv_thin_k" | head -8
done
