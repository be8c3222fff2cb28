mkdir -p gpurun_out/r2b
python -m pytest tests/test_hip_parity.py -m gpu -q -s -p no:cacheprovider --timeout 900 -k "multitile or large_k or memo" > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2b/pytest.log
tail -5 gpurun_out/r2b/pytest.log
for cg in 0 3 7 10 14 17 24; do echo "== cg7 $cg"; python tools/bench_layer.py --layer fc2 --kernel-stats --tune 3=$cg,4=1 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r2b/fc2_cg.log 2>&1
tail -40 gpurun_out/r2b/fc2_cg.log
python tools/bench_layer.py --layer fc2 --kernel-stats --variant 32768 > gpurun_out/r2b/fc2_old.log 2>&1; tail -4 gpurun_out/r2b/fc2_old.log
