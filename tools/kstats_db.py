#!/usr/bin/env python
"""Per-kernel time table from a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace`)."""
import glob
import sqlite3
import sys

for path in ([] if (len(sys.argv) > 1 and sys.argv[1] in ("--busy", "--hist")) else sys.argv[1:]):
    for db in sorted(glob.glob(path) if any(ch in path for ch in "*?") else [path]):
        con = sqlite3.connect(db)
        # KSTATS_TAIL=f: only the kernels that start in the last fraction f of the trace (the steady calibrations of a bench
        # run, without the cold first ones); the conv transposition runs twice per calibration -> calibrations in the window
        import os
        tail = float(os.environ.get("KSTATS_TAIL", "0"))
        t0, t1 = con.execute("select min(start), max(end) from kernels").fetchone()
        cut = t1 - tail * (t1 - t0) if tail > 0 else t0
        rows = con.execute("select name, count(*), avg(end-start), sum(end-start), min(end-start), max(end-start) from kernels "
                           "where start >= ? group by name order by 4 desc", (cut,)).fetchall()
        tot = sum(r[3] for r in rows)
        ncal = sum(r[1] for r in rows if "k_nchw_to_rows" in r[0]) / 2.0
        print(f"== {db}: {tot / 1e6:.2f} ms of kernels" + (f" in the last {tail:.0%} of the trace ({(t1 - cut) / 1e6:.1f} ms of wall-clock, "
              f"{ncal:g} calibrations: {tot / 1e6 / max(ncal, 1):.1f} ms of kernels each)" if tail > 0 else ""))
        for r in rows[:int(os.environ.get("KSTATS_TOP", "14"))]:
            print(f"  {r[0][:84]:84s} n={r[1]:5d} avg={r[2] / 1e3:9.1f} us  {100 * r[3] / tot:5.1f} %")


def busy(db, t_from_first=None):
    """GPU occupancy of a trace: union of the kernel intervals against the wall-clock they span, the idle gaps by the kernel
    that ends before them, and how many kernels overlap on average (python tools/kstats_db.py --busy <db>)."""
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    first = next((r[1] for r in rows if "k_sweep" in r[0]), rows[0][1])      # from the first search kernel on
    rows = [r for r in rows if r[1] >= first]
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    union = 0
    cur_s, cur_e, last_name = rows[0][1], rows[0][2], rows[0][0]
    gaps = {}
    big = []                # the individual gaps: (length, when, kernel before, kernel after)
    for name, s, e in rows[1:]:
        if s > cur_e:
            union += cur_e - cur_s
            gaps[last_name[:60]] = gaps.get(last_name[:60], 0) + (s - cur_e)
            big.append((s - cur_e, cur_e - t0, last_name, name))
            cur_s, cur_e, last_name = s, e, name
        elif e > cur_e:
            cur_e, last_name = e, name
    union += cur_e - cur_s
    ksum = sum(r[2] - r[1] for r in rows)
    print(f"== {db}: wall {1e-6 * (t1 - t0):.1f} ms, some kernel running {1e-6 * union:.1f} ms ({100.0 * union / (t1 - t0):.1f} %), "
          f"sum of kernel times {1e-6 * ksum:.1f} ms (x{ksum / union:.2f} overlap)")
    for k, v in sorted(gaps.items(), key=lambda kv: -kv[1])[:10]:
        print(f"   idle after {k:60s} {1e-6 * v:8.2f} ms")
    import os
    if os.environ.get("KSTATS_GAPS"):      # the longest single gaps of the last calibrations (KSTATS_GAPS=n)
        tail = [g for g in big if g[1] > 0.75 * (t1 - t0)]
        hist = {}
        for g in tail:
            b = 1 << max(0, int(g[0] / 1e3).bit_length())          # power-of-two buckets in us
            hist[b] = (hist.get(b, (0, 0))[0] + 1, hist.get(b, (0, 0))[1] + g[0])
        print("   gaps of the last quarter of the trace by length (us bucket: count, total ms): " +
              ", ".join(f"<{b}: {c}, {1e-6 * v:.1f}" for b, (c, v) in sorted(hist.items())))
        short = lambda n: n.replace("void ", "").replace("p4v::", "").replace("at::native::", "")[:46]
        for g in sorted(tail, key=lambda g: -g[0])[:int(os.environ["KSTATS_GAPS"])]:
            print(f"   {1e-3 * g[0]:8.1f} us at {1e-6 * g[1]:9.2f} ms  after {short(g[2]):46s} before {short(g[3])}")


if len(sys.argv) > 2 and sys.argv[1] == "--busy":
    for path in sys.argv[2:]:
        for db in sorted(glob.glob(path)):
            busy(db)


def hist(db, pattern):
    """Duration histogram of the kernels whose name contains `pattern` (python tools/kstats_db.py --hist <pattern> <db>)."""
    con = sqlite3.connect(db)
    rows = con.execute("select end-start from kernels where name like ?", (f"%{pattern}%",)).fetchall()
    edges = [2, 5, 10, 20, 50, 100, 200, 500, 1000, 5000, 1e9]
    cnt, tot = [0] * len(edges), [0.0] * len(edges)
    for (d,) in rows:
        us = d / 1e3
        for i, e in enumerate(edges):
            if us <= e:
                cnt[i] += 1
                tot[i] += us
                break
    print(f"== {pattern}: {len(rows)} launches, {sum(tot) / 1e3:.1f} ms")
    lo = 0
    for e, c, t in zip(edges, cnt, tot):
        if c:
            print(f"   {lo:>6g} - {e:<6g} us: n={c:6d}  {t / 1e3:8.2f} ms")
        lo = e


if len(sys.argv) > 3 and sys.argv[1] == "--hist":
    for db in sorted(glob.glob(sys.argv[3])):
        hist(db, sys.argv[2])
