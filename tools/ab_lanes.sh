#!/bin/bash
# A/B of capture lanes x hardware queues (GPU_MAX_HW_QUEUES is read at HIP initialisation: one process per setting)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 > /tmp/abl.json
  python - "$label" <<PY
import sys, json
d = json.loads(open("/tmp/abl.json").read())
print(sys.argv[1], round(d["value"], 1), round(d["ms_per_step"], 2), d.get("breakdown"))
PY
}
for spec in "$@"; do
  lanes=${spec%%:*}; q=${spec#*:}
  run "lanes=$lanes queues=$q" P4V_CAPTURE_LANES=$lanes GPU_MAX_HW_QUEUES=$q
done
