#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
bash tools/ab.sh $O/ab42.txt "" "--opt conv.thin_k_blocks=4096" "--opt conv.thin_k_blocks=8192" "--opt conv.thin_k_blocks=1024" "--opt conv.thin_k_blocks=512"
