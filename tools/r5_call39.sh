#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.txt
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/t39_all.txt 2>&1; echo "all rc=$?"; tail -4 $O/t39_all.txt
