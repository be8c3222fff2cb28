import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["breakdown"])
for k,v in d["roofline"]["by_kernel"].items():
    print(k, v["launches"], round(v["ms"],2), {s:(q["launches"], round(q["avg_launch_ms"]*1e3,1), round(q["ms"],2)) for s,q in v["by_stage"].items()})
