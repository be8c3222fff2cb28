#!/bin/bash
# round 2, call q: k_sweep6 ablations on fc1 (timing only)
cd /root/repo; LAYER=${LAYER:-fc1}
mkdir -p gpurun_out
for d in 0 16 32 48; do
  if [ $d = 0 ]; then L=""; else L="P4V_LIB=/root/repo/ptq4vit_amd/csrc/dbg/libp4v_sw6dbg$d.so"; fi
  echo "DBG $d: $(env $L python tools/bench_layer.py --layer $LAYER --rounds 1 --reps 3 --kernel-stats 2>&1 | grep 'sweep6:')"
done | tee gpurun_out/q_ablate.log
