#!/bin/bash
# round 4, run M: ISTFT chunking by rounds + window / interior reciprocal table in registers
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "batch_equals_singles or awkward_lengths or full_size_segment" 2>&1 | tail -5 ) > gpurun_out/r4m_pytest.log
( PB=42 timeout 300 python tools/prof_ops.py r4m_4s 2>&1 | tail -24 ) > gpurun_out/r4m_prof_4s.log
( PB=1 timeout 300 python tools/prof_ops.py r4m_4s_b1 2>&1 | tail -24 ) > gpurun_out/r4m_prof_4s_b1.log
( PB=42 NS=6 timeout 300 python tools/prof_ops.py r4m_6s 2>&1 | tail -24 ) > gpurun_out/r4m_prof_6s.log
tail -4 gpurun_out/r4m_pytest.log; grep -E "^\[|istft|stft" gpurun_out/r4m_prof_4s.log gpurun_out/r4m_prof_4s_b1.log gpurun_out/r4m_prof_6s.log
