cd $GRAFT_REPO_ROOT
echo "== N=2 gloo test mode (two ranks on GPU 0), batch 6"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --batch 6 --backend gloo --no-cpu-baseline --no-roofline 2>&1 | grep -v "amdgpu.ids\|^W0\|^\*\*\*\|Setting OMP" | tail -5
echo "== N=1 default quick (no cpu baseline)"
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -2
