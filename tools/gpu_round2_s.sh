#!/bin/bash
# round 2, call s: cost-model prologue constant of k_sweep6 after the fragment-order prologue
cd /root/repo
for P in 0 150 100 60 30; do
  echo "P6=$P: $(python tools/bench_layer.py --layer qkv,proj,fc1 --rounds 3 --reps 3 --kernel-stats --tune 6=$P 2>&1 | grep 'sweep6:' | sed 's/launches, //; s/TOP.*//' | tr '\n' ' ')"
done
