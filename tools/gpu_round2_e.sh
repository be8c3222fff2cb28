mkdir -p gpurun_out/r2e
python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider --timeout 900 -k "large_k or (multitile and K3072)" 2>&1 | tail -2
for ord in 1 2 3 4; do for cg in 5 10 17; do echo "== order $((ord-1)) cg $cg"; python tools/bench_layer.py --layer fc2 --kernel-stats --tune 3=$cg,5=$ord 2>&1 | grep -E "sweep7 plain|per calib"; done; done
