#!/bin/bash
# repeats the owner-finish engine test under rocgdb until it crashes; prints the backtrace of every thread
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
cat > /tmp/gdbcmds <<'G'
set pagination off
set confirm off
handle SIGSEGV stop print
run
bt 25
info threads
thread apply all bt 12
quit
G
for i in 1 2 3 4; do
  timeout 400 /opt/rocm/bin/rocgdb -q -batch -x /tmp/gdbcmds --args python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "awkward or mono or engine" > gpurun_out/gdb_$i.log 2>&1
  if grep -q "SIGSEGV" gpurun_out/gdb_$i.log; then echo "crash in run $i"; grep -n -A60 "SIGSEGV" gpurun_out/gdb_$i.log | cut -c1-200 | head -120; break; else echo "run $i clean"; tail -3 gpurun_out/gdb_$i.log | cut -c1-200; fi
done
