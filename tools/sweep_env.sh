#!/bin/bash
# usage: [BENCH_ARGS="..."] tools/sweep_env.sh "VAR=val VAR2=val" ...   one bench.py run per argument (A/B inside ONE gpurun
# call: box-to-box variance is +-2 %), prints value and ms/step; picks the JSON line out of noisy stdout (RCCL banners)
for cfg in "$@"; do
  v=$(env $cfg python3 bench.py --steps ${STEPS:-50} --warmup 10 --repeats 3 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "CFG [$cfg] $v"
done
