#!/bin/bash
# kernel timeline of graph-replayed steps: rocprofv3 kernel trace of a short bench run -> gpurun_out/trace/
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace -o tr -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 3 "$@" > $GRAFT_REPO_ROOT/gpurun_out/trace_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/trace_bench.err
