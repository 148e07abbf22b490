#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "norm_finalize_inside or norm_act_fwd_bwd" > $O/t30_ops.txt 2>&1; echo "ops rc=$?"; tail -3 $O/t30_ops.txt
bash tools/ab.sh $O/ab30.txt "GAN_AMD_LIB=gan_amd/libgan_amd_r4.so" "" "GAN_AMD_LIB=gan_amd/libgan_amd_r4.so" ""
