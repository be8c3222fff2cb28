mkdir -p gpurun_out/r2d
python -m pytest tests/test_hip_parity.py -m gpu -q -s -p no:cacheprovider --timeout 900 -k "multitile or large_k" > gpurun_out/r2d/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2d/pytest.log
tail -4 gpurun_out/r2d/pytest.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 4 5; do
  lib=""; [ $v != 0 ] && lib="P4V_LIB=$R/ptq4vit_amd/csrc/dbg/libp4v_sw7dbg$v.so"
  env $lib rocprofv3 --kernel-trace -d $R/gpurun_out/r2d/prof_v$v -o v$v -- python $R/tools/bench_layer.py --layer fc2 --reps 4 > $R/gpurun_out/r2d/v$v.log 2>&1
  python $R/tools/kstats_db.py "$R/gpurun_out/r2d/prof_v$v/*.db" | head -5
  rm -rf $R/gpurun_out/r2d/prof_v$v
done
cd $R
for cg in 0 5 10 17 24; do echo "== cg7 $cg"; python tools/bench_layer.py --layer fc2 --kernel-stats --tune 3=$cg 2>&1 | grep -E "sweep_i8|per calib"; done
