#!/bin/bash
# The other BASELINE.json configurations on one GPU (the headline is the default bench.py): value (layers/s), ms per calibration,
# capture / search split.  tools/other_configs.sh > profiles/rN_other_configs_1gpu.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
one() { # label, bench args...
  label=$1; shift
  timeout 900 python bench.py "$@" --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 > /tmp/oc.json
  python - "$label" <<PY
import sys, json
try:
    d = json.loads(open("/tmp/oc.json").read())
    print(f"{sys.argv[1]:58s} {d['value']:8.1f} layers/s  {d['ms_per_step']:9.1f} ms per calibration  {d.get('breakdown')}")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
one "config 0: deit_tiny/224 BasePTQ W8A8 x4"           --model deit_tiny_patch16_224 --config BasePTQ --calib 4 --steps 10 --warmup 3
one "config 1: vit_small/224 PTQ4ViT W8A8 x32"          --model vit_small_patch16_224 --steps 10 --warmup 3
one "config 2a (headline): vit_base/224 PTQ4ViT W8A8 x32"  --steps 10 --warmup 3
one "config 2b: vit_base/224 PTQ4ViT W6A6 x32"          --bits 6 --steps 10 --warmup 3
one "vit_base/224 BasePTQ (cosine) W8A8 x32"            --config BasePTQ --steps 10 --warmup 3
one "config 3: swin_base/384 PTQ4ViT W8A8 x128"         --model swin_base_patch4_window12_384 --calib 128 --steps 2 --warmup 2
one "config 4: vit_base/384 PTQ4ViT W6A6 x128"          --model vit_base_patch16_384 --bits 6 --calib 128 --steps 2 --warmup 2
