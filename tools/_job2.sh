cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6_cos4; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "cosine or cos" 2>&1 | tail -8
( cd /tmp && P4V_GROUP_CALLS=1 timeout 600 rocprofv3 --kernel-trace -d /tmp/pc -o b -- python $GRAFT_REPO_ROOT/bench.py --config BasePTQ --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > $O/cos_bench.json 2>$O/cos_bench.err )
KSTATS_TOP=16 python tools/kstats_db.py "/tmp/pc/*.db" > $O/cos_kernel_stats.txt
python tools/kstats_grid.py "/tmp/pc/*.db" 30 > $O/cos_kernel_by_grid.txt
head -20 $O/cos_kernel_stats.txt | cut -c1-170
grep "p4v" $O/cos_kernel_by_grid.txt | head -24 | cut -c1-170
