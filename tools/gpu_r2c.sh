#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for st in 1 2; do for b in 12 24; do ( DMX_STREAMS=$st timeout 600 python bench.py --batch $b --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-track --no-single 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('streams $st batch',d['config']['segments_per_gpu_per_step'],'ms/seg',d['config']['ms_per_segment'])" ); done; done
