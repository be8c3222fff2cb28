#!/bin/bash
# round 2, call o: capture graph lanes: tests, then bench at 1/2/4/8 lanes
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_model.py -x -q -m gpu -k "graph or capture" > gpurun_out/o_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/o_tests.log
grep -a "passed\|failed\|rc=" gpurun_out/o_tests.log | tail -3
for L in 1 2 4 8; do
  P4V_CAPTURE_LANES=$L timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 4 > gpurun_out/o_bench_$L.json 2> gpurun_out/o_bench.err
  tail -1 gpurun_out/o_bench_$L.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($L, {k:d[k] for k in (\"value\",\"ms_per_step\",\"breakdown\")})"
done
