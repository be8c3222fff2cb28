#!/bin/bash
# round 2: k_sweep6 weight-search prologue: in place (default) vs fragment-order image (tune 8=1)
cd /root/repo; export TMPDIR=/tmp
for T in 0 1 0 1; do
  echo "EPI6W=$T: $(python tools/bench_layer.py --layer qkv,proj,fc1 --rounds 3 --reps 3 --kernel-stats --tune 8=$T 2>&1 | grep 'sweep6:\|per calibration' | sed 's/launches, //; s/TOP.*//; s/(3 round.*//' | tr '\n' ' ')"
done
