#!/bin/bash
# wall-clock A/B of bench.py under different environments: tools/gpu_bench_ab.sh "A=1" "B=2 C=3" ...
# ("-" = default environment). TESTS=1 runs the GPU parity suite first.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
if [ "${TESTS:-0}" = "1" ]; then timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4; fi
for v in "$@"; do
echo "=== $v"
( [ "$v" != "-" ] && export $v; timeout 600 python bench.py --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --no-roofline --no-single --no-track 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: print(ln.rstrip()); continue
    print(d['value'], 'xRT', d['ms_per_step'], 'ms/step', d['config']['ms_per_segment'], 'ms/seg')
" )
done
