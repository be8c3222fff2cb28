#!/bin/bash
# round 2, call m: k_sweep8 + multi_copy capture: tests, then bench
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_planes.py tests/test_hip_model.py -x -q -m gpu > gpurun_out/m_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/m_tests.log
tail -5 gpurun_out/m_tests.log
timeout 600 python bench.py > gpurun_out/m_bench.json 2> gpurun_out/m_bench.err
cat gpurun_out/m_bench.json
