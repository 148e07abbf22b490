#!/bin/bash
# single-stream rocprofv3 kernel statistics of bench.py variants: tools/prof_ab.sh OUTDIR "<bench args A>" "<bench args B>" ...
# (--no-roofline: the trace holds 25 replays of the captured step + ONE eager warm-up step)
# -> OUTDIR/v<i>_kernel_stats.csv (+ the bench line under the profiler in v<i>.json)
out=$GRAFT_REPO_ROOT/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/v$i -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --sustain 0 --no-roofline --single-stream $v > $out/v$i.json 2> $out/v$i.err
  cp $out/v$i/s_kernel_stats.csv $out/v${i}_kernel_stats.csv
  cp $out/v$i/s_kernel_trace.csv $out/v${i}_kernel_trace.csv
  rm -rf $out/v$i
  i=$((i+1))
done
