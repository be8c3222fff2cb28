#!/usr/bin/env python
"""Join a rocprofv3 profile of `bench.py --profile` with the engine's own per-launch records, launch by launch.

    python tools/prof_join.py --launches L.json --trace "DIR/*.db" [--fetch F.csv] [--write W.csv] [--counters C.csv ...]
                              --out profiles/rNN_production_by_stage

`bench.py --profile --dump-launches L.json` makes every calibration of the process the SAME single-stream production step
(exact pruning + pass memo on), and writes what the engine knows about every sweep launch of one of them, in launch order:
kernel family, stage of the pruned pass (A = all candidates on the sample slice, B1 = the bound, B2 = the survivors, full =
pass not pruned), grid, algorithmic ops and bytes.  With one search stream the kernels of a family appear in the kernel
trace in that same order, once per calibration -- so launch i of a family in the trace IS record i mod n of that family
(grids are checked).  That gives, from the profiler's own clock and counters and for exactly the launches `roofline` in the
bench line describes: launches per calibration, average duration, achieved fraction of the MFMA peak, and -- from the
separate --pmc passes, FETCH_SIZE x 2 (gfx950 counts half of a wide read, MI355X_MICROARCH.md) + WRITE_SIZE -- memory-side
bytes per launch next to the algorithmic bytes.  Output: <out>.json (read back by bench.py as roofline.traffic / .profile)
and <out>.txt (the table).
"""
import argparse
import csv
import glob
import json
import re
import sqlite3
import sys

PEAK_I8, PEAK_F32 = 5000.0, 157.3


def family(name):
    """Kernel family of a demangled kernel name, as the engine's records name it (ptq4vit_amd/_lib.py LAUNCH_KINDS)."""
    n = name.replace("void ", "").replace("p4v::", "")
    m = re.match(r"(k_\w+)<([^>]*)>", n)
    if not m:
        return None
    base, targs = m.group(1), [a.strip() for a in m.group(2).split(",")]
    if base.endswith("_g"):                  # the grouped entry point of the same kernel body (round 6: p4v_calibrate_group)
        base = base[:-2]
    if base == "k_sweep6":
        return "k_sweep6"
    if base == "k_sweep7":
        return "k_sweep7" if targs[0] == "0" else "k_sweep7 (twin)"
    if base in ("k_sweep4", "k_sweep5"):
        return "k_sweep4/5"
    if base == "k_slice_b2":                 # (round 5: B in registers; the engine's records keep the family name)
        return "k_slice_b"
    if base in ("k_sweep9", "k_sweep8", "k_sweep2g", "k_sweep2", "k_sos_split", "k_bound", "k_slice_b", "k_slice_a"):
        return base
    if base == "k_sweep":
        return "k_sweep<float>" if targs[0] == "float" else "k_sweep<int8>"
    return None


def trace_rows(pattern):
    rows = []
    for db in sorted(glob.glob(pattern)):
        con = sqlite3.connect(db)
        cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
        want = [c for c in ("grid_x", "grid_y", "grid_z", "workgroup_x") if c in cols]
        for r in con.execute(f"select name, start, end, {', '.join(want)} from kernels order by start"):
            d = dict(zip(["name", "start", "end"] + want, r))
            f = family(d["name"])
            if f:
                d["family"] = f
                rows.append(d)
    return rows


def counter_rows(path):
    """[(order key, kernel name, {counter: value})] of one rocprofv3 counter-collection csv, one entry per dispatch."""
    per = {}
    with open(path) as fh:
        for r in csv.DictReader(fh):
            key = int(r.get("Dispatch_Id") or r.get("Start_Timestamp") or 0)
            e = per.setdefault(key, {"name": r["Kernel_Name"], "c": {}})
            e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [(k, v["name"], v["c"]) for k, v in sorted(per.items())]


def _source_hash():
    """hash of the engine sources the profile was taken on (bench.py quotes `traffic` only from a profile of the running code)"""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        from ptq4vit_amd import _lib
        return _lib.source_hash()
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rerender", default=None, help="an existing <out>.json: recompute the derived fields and rewrite json + txt (no inputs needed)")
    ap.add_argument("--launches", default=None)
    ap.add_argument("--trace", default=None, help="glob of the rocpd databases of the kernel-trace run")
    ap.add_argument("--fetch", default=None, help="counter csv of the FETCH_SIZE pass")
    ap.add_argument("--write", default=None, help="counter csv of the WRITE_SIZE pass")
    ap.add_argument("--counters", nargs="*", default=[], help="more counter csvs (any counters: per-launch means are reported)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    if a.rerender:
        out = json.load(open(a.rerender))
        derive(out)
        json.dump(out, open(a.rerender, "w"), indent=1)
        render(out, [], a.rerender[:-5] + ".txt")
        print(open(a.rerender[:-5] + ".txt").read())
        return 0
    if not (a.launches and a.trace and a.out):
        ap.error("--launches, --trace and --out are required (or --rerender)")

    L = json.load(open(a.launches))
    recs = L["launches"]
    by_fam = {}
    for r in recs:
        by_fam.setdefault(r["kernel"], []).append(r)

    trace = trace_rows(a.trace)
    t_fam = {}
    for r in trace:
        t_fam.setdefault(r["family"], []).append(r)

    out = {"note": ("rocprofv3 kernel trace + separate --pmc passes of `bench.py --profile` (every calibration = the single-stream "
                    "production step), joined launch by launch with the engine's records (tools/prof_join.py). " + a.note).strip(),
           "model": L.get("model"), "bits": L.get("bits"), "calib": L.get("calib"), "kernels": {}, "join": {},
           "source_hash": _source_hash()}
    lines = []
    for fam, rl in sorted(by_fam.items(), key=lambda kv: -sum(r["ms"] for r in kv[1])):
        tl = t_fam.get(fam, [])
        n = len(rl)
        ok = len(tl) > 0 and len(tl) % n == 0
        ncal = len(tl) // n if ok else 0
        mism = 0
        if ok:
            for i, t in enumerate(tl):
                r = rl[i % n]
                gx = t["grid_x"] // max(1, t.get("workgroup_x", 1))
                if gx != r["grid_x"]:
                    mism += 1
        out["join"][fam] = {"records_per_calibration": n, "trace_launches": len(tl), "calibrations_in_trace": ncal, "grid_mismatches": mism}
        if not ok or mism:
            lines.append(f"!! {fam}: {len(tl)} launches in the trace vs {n} records per calibration ({mism} grid mismatches): not joined")
            continue
        peak = PEAK_F32 if ("float" in fam or "sos" in fam) else PEAK_I8

        def summarise(idx):
            """idx: record indices (within one calibration) to aggregate over all calibrations of the trace"""
            sel = set(idx)
            dur = [t["end"] - t["start"] for i, t in enumerate(tl) if (i % n) in sel]
            alg = sum(rl[i]["alg_ops"] for i in idx)
            iss = sum(rl[i]["ops"] for i in idx)
            byt = sum(rl[i].get("alg_bytes", 0.0) for i in idx)
            ev_ms = sum(rl[i]["ms"] for i in idx)
            avg_ns = sum(dur) / len(dur)
            d = {"launches_per_calibration": len(idx), "avg_launch_ms": avg_ns * 1e-6, "ms_per_calibration": avg_ns * 1e-6 * len(idx),
                 "hip_event_avg_launch_ms": ev_ms / len(idx), "ops_per_launch": alg / len(idx),
                 "achieved": alg / len(idx) / (avg_ns * 1e-9) / 1e12, "issued": iss / len(idx) / (avg_ns * 1e-9) / 1e12,
                 "algorithmic_bytes_per_launch": byt / len(idx)}
            d["frac"] = d["achieved"] / peak
            return d
        k = summarise(range(n))
        k["peak"], k["unit"] = peak, "TOP/s" if peak == PEAK_I8 else "TFLOP/s"
        k["by_stage"] = {}
        for st in ("A", "A2", "B1", "B2", "full"):
            idx = [i for i, r in enumerate(rl) if r["stage"] == st]
            if idx:
                k["by_stage"][st] = summarise(idx)
        out["kernels"][fam] = k

    # counters: the same join on the counter csvs (each pass is its own process with the same launches)
    def join_counters(path, names):
        rows = counter_rows(path)
        cf = {}
        for key, name, c in rows:
            f = family(name)
            if f:
                cf.setdefault(f, []).append(c)
        for fam, cl in cf.items():
            if fam not in out["kernels"]:
                continue
            rl = by_fam[fam]
            n = len(rl)
            if len(cl) % n:
                lines.append(f"!! {fam}: {len(cl)} dispatches in {path} vs {n} records per calibration: counters not joined")
                continue
            k = out["kernels"][fam]
            for cname in sorted({q for c in cl for q in c}):
                if names and cname not in names:
                    continue
                vals = [c.get(cname, 0.0) for c in cl]
                k.setdefault("counters", {})[cname] = {"mean_per_launch": sum(vals) / len(vals), "dispatches": len(vals)}
                for st, sd in k["by_stage"].items():
                    sv = [v for i, v in enumerate(vals) if rl[i % n]["stage"] == st]
                    if sv:
                        sd.setdefault("counters", {})[cname] = sum(sv) / len(sv)
    if a.fetch:
        join_counters(a.fetch, ("FETCH_SIZE",))
    if a.write:
        join_counters(a.write, ("WRITE_SIZE",))
    for path in a.counters:
        join_counters(path, ())
    for fam, k in out["kernels"].items():
        c = k.get("counters", {})
        if "FETCH_SIZE" in c:
            # FETCH_SIZE / WRITE_SIZE are KB; gfx950's FETCH_SIZE counts half of a wide streaming read -> x 2
            k["fetch_bytes_x2_per_launch"] = c["FETCH_SIZE"]["mean_per_launch"] * 1024.0 * 2.0
            k["write_bytes_per_launch"] = c.get("WRITE_SIZE", {"mean_per_launch": 0.0})["mean_per_launch"] * 1024.0
            k["traffic_bytes_per_launch"] = k["fetch_bytes_x2_per_launch"] + k["write_bytes_per_launch"]
            k["traffic_over_algorithmic"] = k["traffic_bytes_per_launch"] / max(1.0, k["algorithmic_bytes_per_launch"])
            for st, sd in k["by_stage"].items():
                sc = sd.get("counters", {})
                if "FETCH_SIZE" in sc:
                    sd["traffic_bytes_per_launch"] = sc["FETCH_SIZE"] * 2048.0 + sc.get("WRITE_SIZE", 0.0) * 1024.0
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            h, m = c["TCC_HIT_sum"]["mean_per_launch"], c["TCC_MISS_sum"]["mean_per_launch"]
            k["l2_hit_rate"] = h / max(1.0, h + m)

    derive(out)
    json.dump(out, open(a.out + ".json", "w"), indent=1)
    render(out, lines, a.out + ".txt")
    print(open(a.out + ".txt").read())


SIMDS, SPEC_GHZ = 1024, 2.4


def derive(out):
    """Matrix-pipe occupancy from the SQ counters, where collected: SQ_VALU_MFMA_BUSY_CYCLES sums the cycles every SIMD's MFMA
    pipe was busy (32 per v_mfma_i32_32x32x32_i8, MI355X_MICROARCH.md) -> busy cycles per SIMD / the launch's cycles at the
    2.4 GHz spec clock (the clock `peak` is quoted at; under MFMA load the part sustains ~1.93 GHz, tools/ubench_mfma.hip:
    4044 of 5033 TOP/s, so the pipe is busy for a larger share of the REAL cycles)."""
    for k in out["kernels"].values():
        for d in [k] + list(k.get("by_stage", {}).values()):
            c = d.get("counters", {})
            v = c.get("SQ_VALU_MFMA_BUSY_CYCLES")
            if v is None:
                continue
            busy = v["mean_per_launch"] if isinstance(v, dict) else v
            d["mfma_busy_of_spec_cycles"] = busy / SIMDS / (d["avg_launch_ms"] * 1e6 * SPEC_GHZ)


def render(out, lines, path):
    with open(path, "w") as fh:
        fh.write(out["note"] + "\n")
        fh.write(f"{'kernel family / stage':28s} {'launches':>8s} {'avg us':>9s} {'(events)':>9s} {'ms/calib':>9s} {'GOP/launch':>11s} {'TOP/s':>8s} {'frac':>6s} "
                 f"{'alg MB':>8s} {'traffic MB':>10s} {'MFMA busy':>9s}\n")
        for fam, k in out["kernels"].items():
            def row(label, d):
                tr = d.get("traffic_bytes_per_launch")
                mb = d.get("mfma_busy_of_spec_cycles")
                fh.write(f"{label:28s} {d['launches_per_calibration']:8d} {d['avg_launch_ms'] * 1e3:9.1f} {d['hip_event_avg_launch_ms'] * 1e3:9.1f} "
                         f"{d['ms_per_calibration']:9.2f} {d['ops_per_launch'] / 1e9:11.1f} {d['achieved']:8.1f} {d['frac']:6.3f} "
                         f"{d['algorithmic_bytes_per_launch'] / 1e6:8.1f} {(tr / 1e6 if tr is not None else float('nan')):10.1f} "
                         f"{(f'{mb:9.3f}' if mb is not None else '        -')}\n")
            row(fam, k)
            for st, sd in k["by_stage"].items():
                row(f"    stage {st}", sd)
        fh.write("MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (launch duration x 2.4 GHz spec clock): occupancy of the matrix pipes, any dtype; '-' = counter not collected\n")
        for fam, j in out["join"].items():
            fh.write(f"join {fam}: {j}\n")
        for ln in lines:
            fh.write(ln + "\n")


if __name__ == "__main__":
    sys.exit(main())
