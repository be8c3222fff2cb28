"""The measurement tooling that bench.py's `roofline` and the multi-GPU prediction rest on, on synthetic inputs (no GPU):
tools/prof_join.py (rocprofv3 kernel trace + PMC passes joined launch by launch with the engine's records) and
tools/predict_scale.py (recomputing the committed prediction from its measured part)."""
import csv
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_prof_join_joins_trace_and_counters_with_the_launch_records(tmp_path):
    recs = []
    # one calibration = 3 k_sweep6 launches (stage A, B1-less: A, A2, B2) and 2 k_bound launches
    for st, gx, ms, ops, byt in (("A", 48, 0.050, 60e9, 10e6), ("A2", 72, 0.100, 240e9, 40e6), ("B2", 900, 0.090, 180e9, 70e6)):
        recs.append({"kernel": "k_sweep6", "stage": st, "grid_x": gx, "grid_z": 1, "ms": ms, "ops": ops * 1.1, "alg_ops": ops, "alg_bytes": byt})
    for gx in (600, 2400):
        recs.append({"kernel": "k_bound", "stage": "B1", "grid_x": gx, "grid_z": 1, "ms": 0.04, "ops": 30e9, "alg_ops": 30e9, "alg_bytes": 150e6})
    lpath = tmp_path / "launches.json"
    json.dump({"model": "m", "bits": 8, "calib": 32, "launches": recs}, open(lpath, "w"))
    db = tmp_path / "t_results.db"
    con = sqlite3.connect(db)
    con.execute("create table kernels (name text, start integer, end integer, grid_x integer, grid_y integer, grid_z integer, workgroup_x integer)")
    t = 0
    fetch_rows, write_rows = [], []
    disp = 0
    for cal in range(4):                               # four identical calibrations in the trace
        for r in recs:
            name = ("void p4v::k_sweep6<0, 12, 2>(p4v::Sweep3Params)" if r["kernel"] == "k_sweep6" else "void p4v::k_bound<0>(p4v::SweepParams)")
            wg = 256
            dur = int(r["ms"] * 1e6 * 0.95)            # the profiler sees 5 % shorter launches than the events
            con.execute("insert into kernels values (?,?,?,?,?,?,?)", (name, t, t + dur, r["grid_x"] * wg, 1, 1, wg))
            t += dur + 5000
            disp += 1
            fetch_rows.append({"Dispatch_Id": disp, "Kernel_Name": name, "Counter_Name": "FETCH_SIZE", "Counter_Value": r["alg_bytes"] / 1024.0})
            write_rows.append({"Dispatch_Id": disp, "Kernel_Name": name, "Counter_Name": "WRITE_SIZE", "Counter_Value": 10.0})
        con.execute("insert into kernels values (?,?,?,?,?,?,?)", ("void at::native::some_torch_kernel()", t, t + 1000, 256, 1, 1, 256))
        t += 2000
    con.commit()
    con.close()
    for path, rows in ((tmp_path / "f.csv", fetch_rows), (tmp_path / "w.csv", write_rows)):
        with open(path, "w", newline="") as fh:
            wr = csv.DictWriter(fh, fieldnames=list(rows[0]))
            wr.writeheader()
            wr.writerows(rows)
    out = tmp_path / "joined"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_join.py"), "--launches", str(lpath), "--trace", str(db),
                    "--fetch", str(tmp_path / "f.csv"), "--write", str(tmp_path / "w.csv"), "--out", str(out)], check=True, capture_output=True)
    d = json.load(open(str(out) + ".json"))
    k6, kb = d["kernels"]["k_sweep6"], d["kernels"]["k_bound"]
    assert d["join"]["k_sweep6"] == {"records_per_calibration": 3, "trace_launches": 12, "calibrations_in_trace": 4, "grid_mismatches": 0}
    assert k6["launches_per_calibration"] == 3 and set(k6["by_stage"]) == {"A", "A2", "B2"}
    assert abs(k6["avg_launch_ms"] - 0.95 * (0.05 + 0.1 + 0.09) / 3) < 1e-6 and abs(k6["hip_event_avg_launch_ms"] - 0.08) < 1e-9
    assert abs(k6["frac"] - (480e9 / 3) / (k6["avg_launch_ms"] * 1e-3) / 1e12 / 5000.0) < 1e-9
    # FETCH_SIZE is KB and counts half of a wide read on gfx950 (x 2), WRITE_SIZE is KB
    assert abs(k6["by_stage"]["B2"]["traffic_bytes_per_launch"] - (2 * 70e6 + 10 * 1024)) < 1.0
    assert abs(kb["traffic_bytes_per_launch"] - (2 * 150e6 + 10 * 1024)) < 1.0 and abs(kb["algorithmic_bytes_per_launch"] - 150e6) < 1.0
    assert "k_sweep6" in open(str(out) + ".txt").read()
    # matrix-pipe occupancy from an SQ counter pass, added to a committed profile without its inputs (--rerender): one MFMA =
    # 32 busy cycles of one SIMD; 100 % of the 1024 SIMDs for the whole launch at the 2.4 GHz spec clock = 1.0
    for d_ in [k6] + list(k6["by_stage"].values()):
        busy = 0.5 * 1024 * d_["avg_launch_ms"] * 1e6 * 2.4
        d_.setdefault("counters", {})["SQ_VALU_MFMA_BUSY_CYCLES"] = {"mean_per_launch": busy, "dispatches": 12} if d_ is k6 else busy
    json.dump(d, open(str(out) + ".json", "w"))
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_join.py"), "--rerender", str(out) + ".json"], check=True, capture_output=True)
    d2 = json.load(open(str(out) + ".json"))
    assert abs(d2["kernels"]["k_sweep6"]["mfma_busy_of_spec_cycles"] - 0.5) < 1e-9
    assert abs(d2["kernels"]["k_sweep6"]["by_stage"]["A2"]["mfma_busy_of_spec_cycles"] - 0.5) < 1e-9
    assert "mfma_busy_of_spec_cycles" not in d2["kernels"]["k_bound"] and "MFMA busy" in open(str(out) + ".txt").read()
    # a trace whose launch count is not a multiple of the records is refused, not silently mis-joined
    con = sqlite3.connect(db)
    con.execute("insert into kernels values (?,?,?,?,?,?,?)", ("void p4v::k_bound<0>(p4v::SweepParams)", t, t + 10, 600 * 256, 1, 1, 256))
    con.commit()
    con.close()
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_join.py"), "--launches", str(lpath), "--trace", str(db), "--out", str(out)],
                   check=True, capture_output=True)
    d2 = json.load(open(str(out) + ".json"))
    assert "k_bound" not in d2["kernels"] and "k_sweep6" in d2["kernels"]


def test_committed_scale_prediction_recomputes_from_its_measurements(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import predict_scale
    res = json.load(open(os.path.join(ROOT, "profiles", "r4_scale_prediction.json")))
    for name, r in res.items():
        p = predict_scale.predict(r["measured"])
        rows = p["rows"]
        assert [q["world"] for q in rows] == [1, 2, 4, 8]
        assert rows[0]["capture_mode"] == "replicated" and abs(rows[0]["speedup"] - 1.0) < 1e-12
        for q, want in zip(rows, r["prediction"]["rows"]):
            assert abs(q["step_s"] - want["step_s"]) < 1e-9 * want["step_s"] and q["capture_mode"] == want["capture_mode"], name
            assert sum(q["modules_per_rank"]) == r["measured"]["modules"] and q["imbalance"] >= 1.0 - 1e-12
        assert all(a["step_s"] > b["step_s"] for a, b in zip(rows, rows[1:])), name           # more GPUs: never slower in the model
    owner, load = predict_scale.lpt({"a": 5.0, "b": 4.0, "c": 3.0, "d": 3.0}, 2)
    assert sorted(load) == [7.0, 8.0] and owner["a"] != owner["b"]
