#!/bin/bash
# round 2, call n: capture passes at a pixel budget: tests, then bench
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_model.py tests/test_hip_configs.py -x -q -m gpu -s > gpurun_out/n_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/n_tests.log
grep -a "\[capture\]\|passed\|failed\|rc=" gpurun_out/n_tests.log | tail
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/n_bench.json 2> gpurun_out/n_bench.err
tail -1 gpurun_out/n_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in (\"value\",\"ms_per_step\",\"breakdown\",\"capture\",\"first_calibration_s\")})"
