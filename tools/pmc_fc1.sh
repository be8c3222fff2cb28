#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pmc_$tag
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python $R/tools/bench_layer.py --layer fc1 --reps 1 > /tmp/pmc_$tag.log 2>&1
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$tag" <<'PY'
import csv, sys, collections
f, tag = sys.argv[1], sys.argv[2]
if not f:
    print(tag, "no output"); sys.exit()
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "k_sweep6" in k or "k_sweep2" in k or "k_pack<signed" in k:
        agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    for c, v in d.items():
        print(f"{tag}: {k:60s} {c:28s} launches {len(v):3d} mean {sum(v)/len(v):.4g} max {max(v):.4g}")
PY
done
