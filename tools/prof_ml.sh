#!/bin/bash
# multi-lane rocprofv3 kernel statistics of bench.py variants (the captured schedule as it runs): tools/prof_ml.sh OUTDIR "<args A>" "<args B>" ...
out=$GRAFT_REPO_ROOT/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/v$i -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --sustain 0 $v > $out/v$i.json 2> $out/v$i.err
  cp $out/v$i/s_kernel_stats.csv $out/v${i}_kernel_stats.csv
  rm -rf $out/v$i
  i=$((i+1))
done
