#!/bin/bash
# A/B of tuning switches on the GROUPED search: step time + k_sweep6 by stage from the roofline records.
#   tools/ab_round6.sh "<tune or -> ..."  [ENV=VALUE ...]      e.g.  tools/ab_round6.sh "- 0=1 6=50" P4V_GROUP_CALLS=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
export GPU_MAX_HW_QUEUES=8
tunes=$1; shift
for kv in "$@"; do export "$kv"; done
for t in $tunes; do
  arg=""; [ "$t" != "-" ] && arg="--tune $t"
  python bench.py $arg --steps 6 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > /tmp/ab6.json
  python - "$t" <<'PY'
import sys, json
d = json.loads(open("/tmp/ab6.json").read())
r = d.get("roofline") or {}
fam = r.get("by_kernel", {})
def st(k):
    v = fam.get(k)
    if not v: return None
    return (v["launches"], round(v["ms"], 2), round(v["frac"], 3), {s: (q["launches"], round(q["avg_launch_ms"] * 1e3, 1), round(q["frac"], 3)) for s, q in v["by_stage"].items()})
print("tune", sys.argv[1], "| step ms", round(d["ms_per_step"], 2), d.get("breakdown"), "| launches", d.get("launches_per_calibration"))
for k in ("k_sweep6", "k_sweep7 (twin)", "k_sweep7", "k_bound"):
    print("    ", k, st(k))
PY
done
