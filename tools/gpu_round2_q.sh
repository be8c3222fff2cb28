#!/bin/bash
cd /root/repo
for i in 1 2 3 4 5 6; do echo "run $i: $(python -m pytest tests/test_hip_parity.py -x -q -m gpu -k 'matmul_baseline_shapes or single_ktile or split_search or matmul_vs' 2>&1 | grep -a '^E  .*Error\|passed\|failed' | head -2 | tr '\n' ' ')"; done
python tools/bench_layer.py --layer qk --rounds 3 --reps 3 --kernel-stats 2>&1 | grep 'sweep_i8\|per calibration'
