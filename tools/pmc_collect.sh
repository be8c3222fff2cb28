#!/bin/bash
# PMC passes for the sweep kernels (MI355X_MICROARCH.md: separate --pmc passes, --kernel-trace only; FETCH_SIZE on gfx950
# counts half of a wide streaming read: doubled by the reader of this file, not here).
#   tools/pmc_collect.sh <layer> <out.json>      e.g. fc1 -> k_sweep6, fc2 -> k_sweep7
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
LAYER=$1; OUT=$2
rm -f /tmp/pmc_rows.txt
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pmc_$tag
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python $R/tools/bench_layer.py --layer $LAYER --reps 1 > /tmp/pmc_$tag.log 2>&1
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" >> /tmp/pmc_rows.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_sweep" in k or "k_pack" in k or "k_prep_epi" in k or "k_sos" in k:
        print(k.replace("\t", " ") + "\t" + r["Counter_Name"] + "\t" + r["Counter_Value"] + "\t" + r.get("Start_Timestamp", "0") + "\t" + r.get("End_Timestamp", "0"))
PY
done
python3 - "$OUT" "$LAYER" <<'PY'
import collections, json, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for line in open("/tmp/pmc_rows.txt"):
    k, c, v, _, _ = line.rstrip("\n").split("\t")
    agg[k][c].append(float(v))
out = {"layer": sys.argv[2], "note": "means per launch over one bench_layer run (warm-up + 1 rep); FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them",
       "kernels": {k: {c: {"launches": len(v), "mean": sum(v) / len(v), "max": max(v)} for c, v in d.items()} for k, d in agg.items()}}
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, d in out["kernels"].items():
    print(k[:70], {c: round(x["mean"], 1) for c, x in d.items()})
PY
