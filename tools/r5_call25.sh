#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
B="python bench.py --no-roofline --no-cpu-baseline --sustain 0 --repeats 3 --steps 50 --warmup 10"
for rep in 1 2; do
for v in 192 256 400 800; do echo "== cyc b1 kb $v"; timeout -k 10 200 $B --model cyclegan --batch 1 --opt conv.own_max_kb=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; done
done
for v in 192 256 400; do echo "== cyc b4 kb $v"; timeout -k 10 200 $B --model cyclegan --batch 4 --opt conv.own_max_kb=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; done
for v in 192 400; do echo "== p2p b1 kb $v"; timeout -k 10 200 $B --batch 1 --opt conv.own_max_kb=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; done
