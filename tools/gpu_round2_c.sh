mkdir -p gpurun_out/r2c
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 4 5 32768; do
  echo "== variant $v"
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2c/prof_v$v -o v$v -- python $GRAFT_REPO_ROOT/tools/bench_layer.py --layer fc2 --variant $v --reps 4 > $GRAFT_REPO_ROOT/gpurun_out/r2c/v$v.log 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/r2c/prof_v$v -name "*kernel_stats.csv" | head -1)
  grep -E "k_sweep|k_pack|k_finish" "$f" | cut -d, -f1-4 | head -8
done
