import sys, numpy as np
d = np.fromfile(sys.argv[1], dtype=np.uint64)
i = 0; k = 0
while i < len(d):
    assert d[i] == 0xABCD
    gx, gz, kt = int(d[i+1]), int(d[i+2]), int(d[i+3]); n = gx * gz
    t = d[i+4:i+4+n*16].reshape(n, 16).astype(np.int64); i += 4 + n*16
    t0 = t[:,0].min()
    start = (t[:,0]-t0)/100.0; pro = (t[:,1]-t[:,0])/100.0; wait0 = (t[:,6]-t[:,1])/100.0
    loop = (t[:,2]-t[:,6])/100.0; tail = (t[:,3]-t[:,2])/100.0; end = (t[:,3]-t0)/100.0
    steps = t[:,7]
    ns_per = loop*1000/np.maximum(steps,1)
    xcc = t[:,5] & 0xf
    print(f"launch {k}: grid {gx}x{gz} ktiles {kt}  kernel span {end.max():.1f} us; WG prologue {pro.mean():.2f} us (max {pro.max():.2f}), first-tile wait {wait0.mean():.2f}, loop {loop.mean():.1f} us, tail {tail.mean():.2f}; "
          f"steps/WG {steps.mean():.0f}; ns per step mean {ns_per.mean():.1f} p10 {np.percentile(ns_per,10):.1f} p90 {np.percentile(ns_per,90):.1f}")
    # occupancy timeline: how many WGs alive over time
    busy = (loop+pro+wait0+tail).sum()
    print(f"    sum WG time {busy:.0f} us = {busy/end.max():.1f} concurrent WGs on average (256 CUs); last start {start.max():.1f} us; "
          f"WGs per XCC: {np.bincount(xcc.astype(int), minlength=8).tolist()}")
    mhz = ((t[:,9]-t[:,8]) / np.maximum((t[:,3]-t[:,0])/100.0, 1e-9))
    print(f'    s_memtime ticks per us: mean {mhz.mean():.1f}')
    k += 1
