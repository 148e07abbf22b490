#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
(echo "== default plans (256-row tiles, one 512-thread workgroup per CU)"; timeout -k 10 120 python tools/probe_gemm_pair.py; echo "== conv.big_tiles=0 (128x128 tiles, 256-thread workgroups, two per CU)"; timeout -k 10 120 python tools/probe_gemm_pair.py --opt conv.big_tiles=0) > $O/probe_gemm_pair.txt 2>&1
cat $O/probe_gemm_pair.txt
timeout -k 10 900 bash tools/make_profiles.sh r05 > $O/make_profiles.log 2>&1; echo "profiles rc=$?"; tail -30 $O/make_profiles.log
