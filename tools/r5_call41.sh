#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
B="python bench.py --no-roofline --no-cpu-baseline --sustain 0 --repeats 3 --steps 50 --warmup 10"
for rep in 1 2; do
for v in 0 1; do for b in 1 4; do echo "== cyc b$b fin $v"; timeout -k 10 200 $B --model cyclegan --batch $b --opt norm.fin_in_apply=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; done; done
done
