#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error" | tail -5 ) > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
( timeout 600 python bench.py 2>&1 | grep '^{' ) > gpurun_out/bench_f.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_f.json'))
c=d['config']
print('value',d['value'],'ms_per_step',d['ms_per_step'],'roofline',d['roofline']['achieved'],d['roofline']['frac'])
print('track_4min',c['track_4min_xRT']['wall_s'],'single',c['single_segment_latency_ms'])
PY
for b in 1 2 4; do ( timeout 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-track 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch',$b,'ms/seg',d['config']['ms_per_segment'],'single',d['config'].get('single_segment_latency_ms'))" ); done
