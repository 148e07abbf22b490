#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd $R
B="python bench.py --no-roofline --no-cpu-baseline --sustain 0 --repeats 3 --steps 50 --warmup 10"
run() { echo "== $1"; shift; timeout -k 10 200 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; }
for rep in 1 2 3; do
run minb64
run minb32 --opt conv.big_min_blocks=32
run minb16 --opt conv.big_min_blocks=16
done > $O/ab24.txt 2>&1
cat $O/ab24.txt
for v in 64 32 16; do echo "== cyc b1 minb $v"; timeout -k 10 200 $B --model cyclegan --batch 1 --opt conv.big_min_blocks=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; done
