#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -q -x -k "column_owner or slab_reduce or finishes_the_layer or layer_stack" > $O/t_own.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/t_own.log
[ $rc -ne 0 ] && exit 1
GAN_AMD_LIB=gan_amd/libgan_amd_diag.so timeout -k 10 120 python tools/diag_own.py conv_fwd 16 2 512 512 2>&1 | grep -v amdgpu.ids
GAN_AMD_LIB=gan_amd/libgan_amd_diag.so timeout -k 10 120 python tools/diag_own.py convT_fwd 16 1 512 512 2>&1 | grep -v amdgpu.ids
B="python bench.py --no-roofline --no-cpu-baseline --sustain 0 --repeats 3 --steps 50 --warmup 10"
run() { echo "== $1"; shift; timeout -k 10 200 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; }
for rep in 1 2 3; do
run own16
run own0 --opt conv.own_max_rows=0
done > $O/ab20.txt 2>&1
cat $O/ab20.txt
for v in 0 16; do echo "== cyc b1 own $v"; timeout -k 10 200 $B --model cyclegan --batch 1 --opt conv.own_max_rows=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; done
