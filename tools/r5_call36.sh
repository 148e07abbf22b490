#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
bash tools/ab.sh $O/ab36.txt "GAN_AMD_LIB=gan_amd/libgan_amd_old.so" "" "GAN_AMD_LIB=gan_amd/libgan_amd_old.so" ""
