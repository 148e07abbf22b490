#!/bin/bash
# One bench.py line per configuration BASELINE.json names (per-GPU shapes) -> gpurun_out/configs_<tag>.jsonl
tag=${1:-r02}
out=gpurun_out/configs_$tag.jsonl
: > $out
run() { python3 bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline "$@" 2>/dev/null >> $out; }
run                                                  # config 2: Pix2Pix 256 bf16 batch 16
run --batch 64
run --dtype f16
run --img-size 512 --batch 8                         # config 4 per-GPU shape
run --model cyclegan --batch 1                       # config 3
run --model cyclegan --batch 4
run --model cyclegan --batch 16
run --model cyclegan --img-size 512 --dtype f16 --batch 16    # config 5 per-GPU shape
python3 - <<PY
import json
for l in open("$out"):
    d = json.loads(l)
    print(f"{d['config']['workload'][:60]:60s} {d['value']:9.1f} {d['unit']:11s} {d['ms_per_step']:8.3f} ms/step")
PY
