#!/bin/bash
# like sweep_env.sh, but picks the JSON line out of noisy stdout (RCCL banners)
for cfg in "$@"; do
  v=$(env $cfg python3 bench.py --steps 30 --warmup 10 --repeats 3 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "CFG [$cfg] $v"
done
