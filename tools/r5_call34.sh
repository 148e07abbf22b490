#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
bash tools/make_profiles.sh r05 > $O/make_profiles.log 2>&1; echo "profiles rc=$?"
cd $R
GAN_AMD_DDP_REHEARSE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --sustain 0 > $O/rehearse.json 2> $O/rehearse.err; echo "rehearse rc=$?"; tail -c 400 $O/rehearse.json
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"; tail -c 300 $O/bench_default.json
