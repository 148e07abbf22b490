#!/bin/bash
# baseline of the re-entered session: default bench line, lane timeline, side-lane marginal costs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_base.json 2> $O/bench_base.err && tail -c 600 $O/bench_base.json
timeout -k 10 200 python tools/lane_timeline.py > $O/timeline.txt 2> $O/timeline.err
timeout -k 10 300 python tools/critpath.py > $O/critpath.txt 2> $O/critpath.err
tail -20 $O/critpath.txt
