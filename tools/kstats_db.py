#!/usr/bin/env python
"""Per-kernel time table from a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace`)."""
import glob
import sqlite3
import sys

for path in sys.argv[1:]:
    for db in sorted(glob.glob(path) if any(ch in path for ch in "*?") else [path]):
        con = sqlite3.connect(db)
        rows = con.execute("select name, count(*), avg(end-start), sum(end-start), min(end-start), max(end-start) from kernels "
                           "group by name order by 4 desc").fetchall()
        tot = sum(r[3] for r in rows)
        print(f"== {db}: {tot / 1e6:.2f} ms of kernels")
        for r in rows[:14]:
            print(f"  {r[0][:84]:84s} n={r[1]:5d} avg={r[2] / 1e3:9.1f} us  {100 * r[3] / tot:5.1f} %")
