#!/bin/bash
# usage: tools/sweep_env.sh "VAR=val VAR2=val" ...   one bench.py run per argument (A/B inside ONE gpurun call), prints img/s
for cfg in "$@"; do
  v=$(env $cfg python3 bench.py --steps 50 --warmup 10 --repeats 3 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "CFG [$cfg] $v"
done
