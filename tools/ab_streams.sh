#!/bin/bash
# A/B of the number of search streams on the default bench (tools/gpu_job.sh style helper): tools/ab_streams.sh "4 3 5 6 4"
cd ${GRAFT_REPO_ROOT:-/root/repo}
for n in $1; do
  python bench.py --search-streams $n --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 > /tmp/ab_$n.json
  python - $n <<PY
import sys, json
d = json.loads(open("/tmp/ab_%s.json" % sys.argv[1]).read())
print("streams", sys.argv[1], round(d["value"], 1), round(d["ms_per_step"], 2), d.get("breakdown"))
PY
done
