#!/bin/bash
# A/B of bench.py variants, interleaved, on one box (rule: compare builds/options inside ONE gpurun call, boxes differ by +-2 %):
#   tools/ab.sh OUT "<bench args A>" "<bench args B>" ...      e.g.  tools/ab.sh gpurun_out/ab.txt "--opt conv.parity_patch=0" ""
# A variant may start with VAR=value words (environment of that run).
out=$1; shift
mkdir -p $(dirname $out)
: > $out
for round in 1 2; do
  for v in "$@"; do
    envs=""; args=""
    for w in $v; do case "$w" in [A-Z_]*=*) envs="$envs $w";; *) args="$args $w";; esac; done
    r=$(env $envs python bench.py --no-cpu-baseline --repeats 3 --sustain 0 $args 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "[$v] round$round: $r" | tee -a $out
  done
done
