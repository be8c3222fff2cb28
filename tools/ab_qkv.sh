cd ${GRAFT_REPO_ROOT:-/root/repo}
export GPU_MAX_HW_QUEUES=8
for v in 0 1 0 1; do
  P4V_QKV_CONTIGUOUS=$v python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 > /tmp/abq.json
  python - "$v" <<PY
import sys, json
d = json.loads(open("/tmp/abq.json").read())
print("qkv_contiguous", sys.argv[1], round(d["value"], 1), round(d["ms_per_step"], 2), d.get("breakdown"))
PY
done
