#!/bin/bash
# round 2, call v: full GPU suite (timed) + bench
cd /root/repo
mkdir -p gpurun_out
SECONDS=0; timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=8 > gpurun_out/v_tests.log 2>&1; echo "Elapsed $SECONDS s" > gpurun_out/v_tests.time
echo "tests rc=$?" >> gpurun_out/v_tests.log
grep -a "passed\|failed\|rc=\|^E \|s call" gpurun_out/v_tests.log | tail -14; grep "Elapsed" gpurun_out/v_tests.time
true
true
