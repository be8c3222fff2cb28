cd ${GRAFT_REPO_ROOT:-/root/repo}
export GPU_MAX_HW_QUEUES=8
run() { # label, env..., -- args
  label=$1; shift
  env "$@" python bench.py $ARGS --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 > /tmp/ab2.json
  python - "$label" <<PY
import sys, json
d = json.loads(open("/tmp/ab2.json").read())
print(sys.argv[1], round(d["value"], 1), round(d["ms_per_step"], 2), d.get("breakdown"))
PY
}
ARGS="" run base X=1
ARGS="--tune 6=130" run P6=13us X=1
ARGS="--tune 6=80" run P6=8us X=1
ARGS="" run streams5 P4V_SEARCH_STREAMS=5
ARGS="" run streams6 P4V_SEARCH_STREAMS=6
ARGS="" run streams3 P4V_SEARCH_STREAMS=3
ARGS="" run lanes2 P4V_CAPTURE_LANES=2
ARGS="" run lanes4 P4V_CAPTURE_LANES=4
ARGS="" run base2 X=1
