#!/bin/bash
# round 2: attention sweeps (k_sweep9 / k_sweep8 / k_sweep2 / k_sos_split): parity + layer bench
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "attention or matmul or single_ktile or split_search or block16" > gpurun_out/t_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/t_tests.log
grep -a "passed\|failed\|rc=\|^E " gpurun_out/t_tests.log | tail -6
for V in 524288 1048576; do echo "variant $V: $(python tools/bench_layer.py --layer qk --rounds 3 --reps 3 --kernel-stats --variant $V 2>&1 | grep 'sweep_i8\|per calibration' | sed 's/TOP.*//; s/(3 round.*//' | tr '\n' ' ')"; done
python tools/module_times.py swin_base_patch4_window12_384 16 2>&1 | grep "matmul1\|total"
