#!/bin/bash
# round 2, call r: k_sweep6 with fragment-order prologue: parity tests, layer bench, bench
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_granular.py tests/test_hip_model.py -x -q -m gpu > gpurun_out/r_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r_tests.log
grep -a "passed\|failed\|rc=" gpurun_out/r_tests.log | tail -3
python tools/bench_layer.py --layer qkv,proj,fc1 --rounds 3 --reps 3 --kernel-stats 2>&1 | grep "sweep6:\|per calibration"
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r_bench.json 2> gpurun_out/r_bench.err
tail -1 gpurun_out/r_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in (\"value\",\"ms_per_step\",\"breakdown\")}); r=d['roofline']; print(r['frac'], r['avg_launch_ms'], r['all_int8_sweeps']['frac'])"
