#!/bin/bash
# dead-tap / zero-gradient skip: op test, A/B; golden tests on regenerated fixtures; in-kernel breakdown of the parity-patch launches
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_golden_full.py tests/test_gpu_ops.py -k "golden or oracle_values or fused_adam" -q -s > $O/t_new3.log 2>&1; echo "pytest rc=$?" >> $O/t_new3.log
grep -n "passed\|failed\|pytest rc" $O/t_new3.log
B="python bench.py --no-roofline --no-cpu-baseline --sustain 0 --repeats 3 --steps 50 --warmup 10"
run() { echo "== $1"; shift; timeout -k 10 200 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; }
for rep in 1 2; do
run dead1
run dead0 --opt wgrad.dead_taps=0
run noalt "--step-attr=wgrad_alt=()"
run cyc_b1_dead1 --model cyclegan --batch 1
run cyc_b1_dead0 --model cyclegan --batch 1 --opt wgrad.dead_taps=0
done > $O/ab_dead.txt 2>&1
cat $O/ab_dead.txt
export GAN_AMD_LIB=gan_amd/libgan_amd_diag.so
(python tools/diag_gemm.py convT_fwd 16 64 256 64 2; STATS=1 python tools/diag_gemm.py convT_fwd 16 64 256 64 2; python tools/diag_gemm.py conv_dgrad 16 64 128 64 2; python tools/diag_gemm.py conv_dgrad 32 64 128 64 2; python tools/diag_gemm.py conv_fwd 16 128 64 128 2;  python tools/diag_gemm.py conv_fwd 32 128 64 128 2; python tools/diag_gemm.py conv_fwd 16 64 128 256 2) > $O/diag_par.txt 2>&1
cat $O/diag_par.txt
