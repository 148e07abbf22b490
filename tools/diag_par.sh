#!/bin/bash
# in-kernel time breakdown of the parity-patch kernel (diagnostic build with stamps): up6 forward and down1 dgrad at batch 16
set -e
tools/diag_build.sh > /dev/null 2>&1
export GAN_AMD_LIB=gan_amd/libgan_amd_diag.so
python tools/diag_gemm.py convT_fwd 16 64 256 64 2
STATS=1 python tools/diag_gemm.py convT_fwd 16 64 256 64 2
python tools/diag_gemm.py conv_dgrad 16 64 128 64 2
