#!/bin/bash
cd /root/repo
for S in 2 3 4 6; do
  P4V_SEARCH_STREAMS=$S timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 5 > /tmp/xb.json 2>/dev/null
  tail -1 /tmp/xb.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($S, round(d['value'],1), round(d['ms_per_step'],1), d['breakdown'])"
done
