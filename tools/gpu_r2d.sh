#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "reduced_segment or full_size_segment or batch_equals or awkward or deterministic_inputs or track_vs_oracle or stream_schedule or graph" 2>&1 | tail -15 ) > gpurun_out/pytest_quick.log
cat gpurun_out/pytest_quick.log
( PB=1 REPS=10 timeout 300 python tools/prof_ops.py b1 2>&1 | grep -v amdgpu.ids ) > gpurun_out/prof_b1.log
( PB=4 REPS=5 timeout 300 python tools/prof_ops.py b4 2>&1 | grep -v amdgpu.ids ) > gpurun_out/prof_b4.log
( PB=24 REPS=3 timeout 300 python tools/prof_ops.py b24 2>&1 | grep -v amdgpu.ids | head -3 ) > gpurun_out/prof_b24.log
cat gpurun_out/prof_b1.log gpurun_out/prof_b4.log gpurun_out/prof_b24.log
for b in 1 2 4; do ( timeout 600 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-track 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('batch',d['config']['segments_per_gpu_per_step'],'ms/seg',d['config']['ms_per_segment'],'single',d['config']['single_segment_latency_ms'])" ); done
