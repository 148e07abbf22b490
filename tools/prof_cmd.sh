#!/bin/bash
# usage: tools/prof_cmd.sh <tag> <python script + args...> : rocprofv3 kernel-trace stats of one command -> gpurun_out/<tag>_stats.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o p -- python3 "$@" > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/prof_$tag/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us avg  x{r['Calls']:>4}  {r['Name'][:90]}")
PY
