#!/bin/bash
# round 2: matmul sweeps (k_sweep8 / k_sweep2 / k_sos_split): parity + layer bench
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_planes.py tests/test_hip_granular.py -x -q -m gpu > gpurun_out/t_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/t_tests.log
grep -a "passed\|failed\|rc=\|^E " gpurun_out/t_tests.log | tail -8
python tools/bench_layer.py --layer qk,sv --rounds 3 --reps 3 --kernel-stats 2>&1 | grep 'sweep_i8\|sweep_f32\|per calibration'
