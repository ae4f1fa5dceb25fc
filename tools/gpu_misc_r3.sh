#!/bin/bash
# round-3 side measurements: two streams at the bench batch (the product forks only below 8 segments), the v3 bench in the
# N=2 gloo test mode, the full GPU suite with its summary kept
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
Q="--steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-single --no-track --no-split-probe"
for st in 1 2; do
  echo "== DMX_STREAMS=$st, batch 42 / 12"
  DMX_STREAMS=$st timeout 600 python bench.py $Q 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  DMX_STREAMS=$st timeout 600 python bench.py $Q --batch 12 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
echo "== v3, N=2 gloo test mode (two ranks on GPU 0), batch 6"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --model v3 --gpus 2 --steps 2 --warmup 1 --batch 6 --backend gloo --no-cpu-baseline --no-roofline 2>&1 | grep -E "test mode|^\{" | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/suite.txt 2>&1; grep -E "passed|failed" gpurun_out/suite.txt | tail -3
