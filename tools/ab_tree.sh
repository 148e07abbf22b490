#!/bin/bash
# A/B of two source trees on one box: the working tree against _ab/base (e.g. `git archive HEAD | tar -x -C _ab/base` plus its
# built libgan_amd.so); interleaved bench.py runs.   tools/ab_tree.sh OUT [bench args]
out=$1; shift
mkdir -p $(dirname $out); : > $out
for round in 1 2 3; do
  for tree in _ab/base .; do
    r=$(cd $tree && python bench.py --no-cpu-baseline --repeats 3 --sustain 0 "$@" 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "[$tree] round$round: $r" | tee -a $out
  done
done
