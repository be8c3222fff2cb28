mkdir -p gpurun_out/r2h
python -m pytest tests -m gpu -q -s -p no:cacheprovider --timeout 900 > gpurun_out/r2h/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2h/pytest.log
tail -4 gpurun_out/r2h/pytest.log
for l in l_qkv l_proj l_fc1 l_fc2 s_fc2; do for v in 0 32768; do echo "== $l variant $v"; python tools/bench_layer.py --layer $l --variant $v 2>&1 | grep -E "per calib"; done; done
python bench.py --steps 5 --warmup 2 > gpurun_out/r2h/bench.json 2> gpurun_out/r2h/bench.err; python - <<'PY'
import json
l=[x for x in open("gpurun_out/r2h/bench.json") if x.startswith("{")][-1]
d=json.loads(l); print(d["value"], d["ms_per_step"], d["breakdown"], d["roofline"]["all_int8_sweeps"] if d["roofline"] else None)
PY
