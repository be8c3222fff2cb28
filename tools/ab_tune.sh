#!/bin/bash
# A/B of p4v_debug_set_tuning switches on the default bench: tools/ab_tune.sh "<tune or -> ..."   e.g.  tools/ab_tune.sh "- 12=8 -"
cd ${GRAFT_REPO_ROOT:-/root/repo}
i=0
for t in $1; do
  i=$((i+1)); arg=""; [ "$t" != "-" ] && arg="--tune $t"
  python bench.py $arg --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 > /tmp/abt_$i.json
  python - "$t" /tmp/abt_$i.json <<PY
import sys, json
d = json.loads(open(sys.argv[2]).read())
print("tune", sys.argv[1], round(d["value"], 1), round(d["ms_per_step"], 2), d.get("breakdown"))
PY
done
