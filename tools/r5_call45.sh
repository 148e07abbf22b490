#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "finishing_reduce_on_512 or split_k_layer_finished or split_k_dgrad_finishes" > $O/t45_ops.txt 2>&1; echo "ops rc=$?"; tail -5 $O/t45_ops.txt
bash tools/ab.sh $O/ab45.txt "" "--opt conv.skn512_min_rows=513" "--opt conv.skn512_min_rows=257"
B="python bench.py --no-roofline --no-cpu-baseline --sustain 0 --repeats 3 --steps 50 --warmup 10"
for rep in 1 2; do for v in 0 513 257; do for b in 1 4; do echo "== cyc b$b skn512 $v"; timeout -k 10 200 $B --model cyclegan --batch $b --opt conv.skn512_min_rows=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step_all_repeats'])"; done; done; done
