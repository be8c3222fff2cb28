mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -q -s -p no:cacheprovider --timeout 900 > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
python bench.py --steps 5 --warmup 2 > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err; tail -c 600 gpurun_out/r2a/bench.json
python tools/bench_layer.py --layer qkv,proj,fc1,fc2,qk,sv --kernel-stats > gpurun_out/r2a/layers.log 2>&1; tail -20 gpurun_out/r2a/layers.log
