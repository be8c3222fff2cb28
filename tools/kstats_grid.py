"""Kernel time grouped by (kernel, grid): which launch shapes the time goes to.  python tools/kstats_grid.py <db glob> [top]"""
import glob
import sqlite3
import sys

for db in ([] if (len(sys.argv) > 2 and sys.argv[2] == "ctx") else sorted(glob.glob(sys.argv[1]))):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    g = [c for c in cols if c.startswith("grid")]
    w = [c for c in cols if c.startswith("workgroup")]
    if not g:
        print("columns:", cols)
        continue
    q = f"select name, {','.join(g + w)}, count(*), sum(end-start) from kernels group by name, {','.join(g + w)} order by 0+sum(end-start) desc"
    rows = con.execute(q).fetchall()
    total = sum(r[-1] for r in rows)
    print(f"== {db}: {total / 1e6:.1f} ms; columns {g + w}")
    for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
        name = r[0].replace("p4v::", "").replace("void ", "")[:44]
        print(f"  {name:44s} {str(r[1:-2]):34s} n={r[-2]:5d} avg={r[-1] / r[-2] / 1e3:8.1f} us  {100.0 * r[-1] / total:5.1f} %")


def context(db, name_like, grid_x, grid_y, n=6):
    """What runs right after the launches of one shape (python tools/kstats_grid.py <db> ctx <name> <grid_x> <grid_y>)."""
    con = sqlite3.connect(db)
    rows = con.execute("select name, grid_x, grid_y, start, end from kernels order by start").fetchall()
    seen = {}
    for i, r in enumerate(rows):
        if name_like in r[0] and r[1] == grid_x and r[2] == grid_y:
            key = " | ".join(f"{q[0].replace('p4v::', '').replace('void ', '')[:28]}({q[1]},{q[2]})" for q in rows[max(0, i - 2): i + n])
            seen[key] = seen.get(key, 0) + 1
    for k, v in sorted(seen.items(), key=lambda kv: -kv[1])[:8]:
        print(f"  x{v}: {k}")


if len(sys.argv) > 5 and sys.argv[2] == "ctx":
    for db in sorted(glob.glob(sys.argv[1])):
        context(db, sys.argv[3], int(sys.argv[4]), int(sys.argv[5]))
