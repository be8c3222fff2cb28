python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider --timeout 900 -k "large_k or (multitile and K3072) or (multitile and K1024)" 2>&1 | tail -2
R=$GRAFT_REPO_ROOT
for v in 0 4 5; do
  lib=""; [ $v != 0 ] && lib="P4V_LIB=$R/ptq4vit_amd/csrc/dbg/libp4v_sw7dbg$v.so"
  echo "== dbg $v"; env $lib python tools/bench_layer.py --layer fc2 --kernel-stats 2>&1 | grep -E "sweep7 plain|per calib"
done
