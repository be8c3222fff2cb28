mkdir -p gpurun_out/r2k
python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 -x 2>&1 | tail -3
python bench.py --steps 8 --warmup 3 > gpurun_out/r2k/bench.json 2> gpurun_out/r2k/bench.err; python - <<'PY'
import json
l=[x for x in open("gpurun_out/r2k/bench.json") if x.startswith("{")][-1]
d=json.loads(l); print(d["value"], d["ms_per_step"], d["breakdown"], d["first_calibration_s"]); r=d["roofline"]; print({k: (round(v["ms"],1), round(v["frac"],3)) for k,v in r["by_kernel"].items() if v}, r["all_int8_sweeps"]["frac"], r["traffic"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["cpu_model"])
PY
cd /tmp && export TMPDIR=/tmp && P4V_SEARCH_STREAMS=1 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r2k/prof_bench -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/kstats_db.py "$GRAFT_REPO_ROOT/gpurun_out/r2k/prof_bench/*.db" | tee $GRAFT_REPO_ROOT/gpurun_out/r2k/bench_1stream_kernel_stats.txt | head -24
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r2k/prof_bench
