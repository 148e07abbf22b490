#!/bin/bash
# A/B of bench.py under environment variants, interleaved, on one box:  tools/ab.sh OUT "VAR=a" "VAR=b" ...
out=$1; shift
mkdir -p $(dirname $out)
: > $out
for round in 1 2; do
  for v in "$@"; do
    r=$(env $v python bench.py --no-cpu-baseline --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "$v round$round: $r" | tee -a $out
  done
done
