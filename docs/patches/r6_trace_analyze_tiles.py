"""Per-workgroup timestamps of the stationary sweeps (tuning builds: -DP4V_TRACE, $P4V_TRACE_FILE): where a k_sweep6 workgroup's
time goes -- slab prologue (stationary fragments), tile prologues (epilogue operands + ring warm-up), candidate loops, tails.
  python tools/trace_analyze.py <trace file> [min workgroups]"""
import sys, numpy as np
d = np.fromfile(sys.argv[1], dtype=np.uint64)
min_wg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
i = 0; k = 0
while i < len(d):
    assert d[i] == 0xABCD
    gx, gz, kt = int(d[i+1]), int(d[i+2]), int(d[i+3]); n = gx * gz
    t = d[i+4:i+4+n*16].reshape(n, 16).astype(np.int64); i += 4 + n*16
    k += 1
    if n < min_wg:
        continue
    live = t[:, 10] > 0
    t0 = t[:, 0].min()
    span = (t[:, 3].max() - t0) / 100.0
    if not live.any():
        print(f"launch {k-1}: grid {gx}x{gz}: no workgroup ran a tile (span {span:.1f} us)")
        continue
    L = t[live]
    slab = L[:, 6] / 100.0; pro = L[:, 1] / 100.0; loop = L[:, 2] / 100.0; tail = L[:, 4] / 100.0
    total = (L[:, 3] - L[:, 0]) / 100.0
    tiles = L[:, 10]; steps = L[:, 7]
    print(f"launch {k-1}: grid {gx}x{gz} ktiles {kt}: span {span:.1f} us; {int(live.sum())} workgroups ran {tiles.mean():.1f} tiles of {steps.sum()/max(1,tiles.sum()):.1f} candidates; "
          f"per workgroup: total {total.mean():.1f} us = slab prologue {slab.mean():.2f} + tile prologues {pro.mean():.2f} ({(pro/tiles).mean():.2f} per tile) + "
          f"loops {loop.mean():.1f} ({1000*(loop/np.maximum(steps,1)).mean():.0f} ns per candidate) + tails {tail.mean():.2f} ({(tail/tiles).mean():.2f} per tile)")
    print(f"    sum of workgroup time {total.sum():.0f} us = {total.sum()/span:.1f} workgroups resident on average (256 CUs); loops are {100*loop.sum()/total.sum():.0f} % of it; "
          f"last start {((L[:,0]-t0)/100.0).max():.1f} us")
