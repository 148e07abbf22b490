#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -k 10 200 python bench.py > $O/bench_default_final2.json 2> $O/bench_default_final2.err; echo "bench rc=$?"
timeout -k 10 900 bash tools/make_profiles.sh r05 > $O/make_profiles2.log 2>&1; echo "profiles rc=$?"; tail -3 $O/make_profiles2.log
