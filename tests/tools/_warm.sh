#!/bin/bash
out=gpurun_out/r06warm; mkdir -p $out
run() { timeout 150 python bench.py --gpus 1 --steps 20 --warmup $1 --no-cpu --no-extras --no-latency 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('warmup $1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do run 5; done
for i in 1 2 3; do run 405; run 5; done
