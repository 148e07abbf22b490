#!/bin/bash
# Diagnostic build of the kernel library with in-kernel stamps (-DGAN_DIAG) -> gan_amd/libgan_amd_diag.so.
# The product library never contains the stamps; select this one with GAN_AMD_LIB=gan_amd/libgan_amd_diag.so.
set -e
cd "$(dirname "$0")/../gan_amd/csrc"
mkdir -p /tmp/gan_diag
for f in conv_gemm conv_own thin wgrad norm elementwise; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DGAN_DIAG -Wno-unused-variable -Wno-unused-function -c $f.hip -o /tmp/gan_diag/$f.o &
done
wait
g++ -O2 -std=c++17 -fPIC -msse4.2 -c host_util.cpp -o /tmp/gan_diag/host_util.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/gan_diag/*.o -o ../libgan_amd_diag.so
