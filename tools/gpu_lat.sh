#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lat
timeout 600 rocprofv3 --kernel-trace -d /tmp/lat -o lat -- python $R/tools/latency_timeline.py run 2>&1 | grep -v amdgpu.ids | tail -3
db=$(find /tmp/lat -name "*.db" | head -1)
cd $R
python tools/latency_timeline.py analyse $db ${1:-0}
