#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
( DMX_LIB=$R/demucs_cpp_amd/lib/libdemucs_hip_timing.so timeout 300 python tools/gpu_wg_timeline.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r4l_wg_timeline2.log
grep -v "workgroups in their\|start times\|wave [123]" gpurun_out/r4l_wg_timeline2.log
( PB=42 timeout 300 python tools/prof_ops.py r4l_4s 2>&1 | tail -24 ) > gpurun_out/r4l_prof_4s.log
( PB=42 DMX_GEMM=f32 timeout 300 python tools/prof_ops.py r4l_4s_f32 2>&1 | tail -24 ) > gpurun_out/r4l_prof_4s_f32.log
head -8 gpurun_out/r4l_prof_4s.log; head -8 gpurun_out/r4l_prof_4s_f32.log
for mode in 0 1; do
  ( DMX_SPLIT_LIN=$mode timeout 300 python tools/gpu_lin_ab.py run /tmp/lin_$mode.npz 2>&1 | tail -3 ) > gpurun_out/r4l_ab_$mode.log
done
( python tools/gpu_lin_ab.py cmp /tmp/lin_0.npz /tmp/lin_1.npz ) > gpurun_out/r4l_cmp.log 2>&1; cat gpurun_out/r4l_ab_*.log gpurun_out/r4l_cmp.log | grep -v amdgpu.ids
( timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 ) > gpurun_out/r4l_bench_4s.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4l_bench_4s.json")); c=d["config"]; print("bench 4s", d["value"], d["ms_per_step"], c.get("f32_mfma_xRT"), d["roofline"]["kernel"], d["roofline"]["achieved"], d["roofline"]["frac"])
PY
