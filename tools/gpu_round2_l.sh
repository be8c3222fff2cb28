python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider --timeout 900 -k "matmul or single_ktile" 2>&1 | tail -3
for v in 0 65536; do echo "== variant $v"; python tools/bench_layer.py --layer qk --kernel-stats --variant $v 2>&1 | grep -E "sweep_i8|per calib"; done
