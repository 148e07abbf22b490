#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv2d_fwd_dgrad_wgrad or convT2d_fwd_dgrad_wgrad or thin" > $O/t44_ops.txt 2>&1; echo "ops rc=$?"; tail -3 $O/t44_ops.txt
bash tools/ab.sh $O/ab44.txt "GAN_AMD_LIB=gan_amd/libgan_amd_prev.so" "" "GAN_AMD_LIB=gan_amd/libgan_amd_prev.so" ""
