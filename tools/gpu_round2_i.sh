mkdir -p gpurun_out/r2i
python -m pytest tests/test_hip_model.py -m gpu -q -s -p no:cacheprovider --timeout 900 -k "cached_capture or graph_capture or repeated or end_to_end" 2>&1 | tail -3
bash tools/pmc_collect.sh fc1 $GRAFT_REPO_ROOT/gpurun_out/r2i/r2_pmc_fc1_sweep6.json > gpurun_out/r2i/pmc_fc1.log 2>&1; tail -3 gpurun_out/r2i/pmc_fc1.log
bash tools/pmc_collect.sh fc2 $GRAFT_REPO_ROOT/gpurun_out/r2i/r2_pmc_fc2_sweep7.json > gpurun_out/r2i/pmc_fc2.log 2>&1; tail -4 gpurun_out/r2i/pmc_fc2.log
cd $GRAFT_REPO_ROOT
python bench.py --steps 5 --warmup 3 > gpurun_out/r2i/bench.json 2> gpurun_out/r2i/bench.err; python - <<'PY'
import json
l=[x for x in open("gpurun_out/r2i/bench.json") if x.startswith("{")][-1]
d=json.loads(l); print(d["value"], d["ms_per_step"], d["breakdown"], d["first_calibration_s"]); r=d["roofline"]; print({k: (round(v["ms"],1), round(v["frac"],3)) for k,v in r["by_kernel"].items() if v}, r["all_int8_sweeps"]["frac"], r["kernel"][:10])
PY
cd /tmp && rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r2i/prof_bench -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/kstats_db.py "$GRAFT_REPO_ROOT/gpurun_out/r2i/prof_bench/*.db" | tee $GRAFT_REPO_ROOT/gpurun_out/r2i/bench_kernel_stats.txt | head -16
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r2i/prof_bench
