#!/usr/bin/env python
"""ISA-level report of the HIP kernels (tuning aid, CPU only: hipcc cross-compiles gfx950).

    python tools/isa_report.py                      # registers / scratch / occupancy of every kernel
    python tools/isa_report.py --gaps k_sweep6ILi0ELi12ELi2   # instructions between consecutive MFMAs in its hot loop

Compiles ptq4vit_amd/csrc/p4v_api.hip with the flags of __graft_entry__.build() plus -save-temps into a scratch
directory and reads the generated assembly.  The "gaps" view is what DESIGN.md s5.1 argues with: a wave that is alone
on its SIMD hides about five single-issue instructions per 32x32x32 MFMA.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-shared", "-Wno-unused-value",
         "-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"]


def compile_asm(workdir):
    src = os.path.join(ROOT, "ptq4vit_amd", "csrc", "p4v_api.hip")
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + ["-save-temps=obj", src, "-o", os.path.join(workdir, "lib.so")]
    subprocess.run(cmd, cwd=workdir, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return os.path.join(workdir, "p4v_api-hip-amdgcn-amd-amdhsa-gfx950.s")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def resources(lines):
    res, name = {}, None
    for l in lines:
        m = re.match(r"^(_ZN3p4v\S+):", l)
        if m:
            name = m.group(1)
        m = re.match(r"^; (NumVgprs|NumAgprs|ScratchSize|Occupancy): (\d+)", l)
        if m and name:
            res.setdefault(name, {})[m.group(1)] = int(m.group(2))
    return res


def gaps(lines, sym):
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3p4v") and sym in l and l.rstrip().endswith(":") is False and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur, label = [], [], "entry"
    for l in lines[start:end]:
        s = l.strip()
        if re.match(r"^\.LBB\d+_\d+:", s):
            blocks.append((label, cur))
            cur, label = [], s.split(":")[0]
            continue
        if not s or s[0] in ";.":
            continue
        cur.append(s.split()[0])
    blocks.append((label, cur))
    label, seq = max(blocks, key=lambda b: sum("mfma" in o for o in b[1]))
    g, run = [], 0
    for o in seq:
        if "mfma" in o:
            g.append(run)
            run = 0
        else:
            run += 1
    g.append(run)
    n = sum("mfma" in o for o in seq)
    print(f"{label}: {len(seq)} instructions, {n} MFMAs, {(len(seq) - n) / max(1, n):.2f} others per MFMA")
    print("instructions between consecutive MFMAs:", g)
    mix = {}
    for o in seq:
        mix[o] = mix.get(o, 0) + 1
    print("mix:", sorted(mix.items(), key=lambda kv: -kv[1])[:14])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaps", metavar="SUBSTR", help="mangled-name substring of one kernel, e.g. k_sweep6ILi0ELi12ELi2")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as d:
        lines = open(compile_asm(d)).read().split("\n")
    if a.gaps:
        gaps(lines, a.gaps)
        return
    res = resources(lines)
    names = demangle(list(res))
    print(f"{'kernel':72s} {'VGPR':>5s} {'AGPR':>5s} {'scratch':>8s} {'waves/SIMD':>10s}")
    for k, r in res.items():
        flag = "  <-- scratch" if r.get("ScratchSize", 0) else ""
        print(f"{names[k][:72]:72s} {r.get('NumVgprs', 0):5d} {r.get('NumAgprs', 0):5d} {r.get('ScratchSize', 0):8d} {r.get('Occupancy', 0):10d}{flag}")


if __name__ == "__main__":
    sys.exit(main())
