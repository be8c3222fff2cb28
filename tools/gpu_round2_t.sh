#!/bin/bash
# round 2, call t: k_sos_split: A/B + oracle parity, layer bench
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "split_search or matmul or attention" > gpurun_out/t_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/t_tests.log
grep -a "passed\|failed\|rc=\|^E " gpurun_out/t_tests.log | tail -8
for V in 0 131072; do echo "variant $V: $(python tools/bench_layer.py --layer sv --rounds 3 --reps 3 --kernel-stats --variant $V 2>&1 | grep 'sweep_f32\|per calibration' | tr '\n' ' ')"; done
