"""Shared parity helpers for oracle / HIP tests (tie-aware argmax rule)."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Floating-point tolerance of the parity bar.  Scores are float32 sums of ~1e3..1e7 terms computed
# with different summation orders (torch CPU kernels / numpy / MFMA + wave reductions); the
# reference's own GEMM rounding noise is of the same size.  BASELINE.json asks for scaling factors
# within +-0.1 %: an index match makes them bit-identical, a near-tie moves them by one grid step.
SCORE_RTOL = 2e-4
TIE_RTOL = 1e-5          # a differing selection must be a tie to <= 1e-5 relative by the ORACLE's own scores


# ---- margins: how far from the bar each test actually is ------------------------------------------------------------------
# Every assertion below also records the worst value it saw, keyed by the running test; the table is written when the process
# ends ($P4V_MARGINS_OUT, default gpurun_out/parity_margins.json when a GPU is present) and committed as
# profiles/r6_parity_margins.json: score error against SCORE_RTOL, oracle gap of every differing selection against TIE_RTOL,
# candidate-table steps between differing intervals.
_MARGINS = {}


def record_margin(kind, value, extra=None):
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    e = _MARGINS.setdefault(test, {})
    if value is not None and (kind not in e or value > e[kind]):
        e[kind] = float(value)
    if extra:
        for k, v in extra.items():
            e[k] = e.get(k, 0) + v


def _dump_margins():
    if not _MARGINS:
        return
    path = os.environ.get("P4V_MARGINS_OUT")
    if not path:
        try:
            import torch
            if not torch.cuda.is_available():
                return
        except Exception:      # noqa: BLE001
            return
        path = os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out", "parity_margins.json")
    path = os.path.abspath(path)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    old = {}
    if os.path.exists(path):
        try:
            old = json.load(open(path))
        except Exception:      # noqa: BLE001
            old = {}
    old.update(_MARGINS)
    with open(path, "w") as fh:
        json.dump({"bars": {"SCORE_RTOL": SCORE_RTOL, "TIE_RTOL": TIE_RTOL, "GRID_TOL": GRID_TOL, "CAPTURE_TOL": CAPTURE_TOL, "max_grid_steps": MAX_GRID_STEPS},
                   **{k: v for k, v in sorted(old.items()) if k != "bars"}}, fh, indent=1)


import atexit  # noqa: E402
atexit.register(_dump_margins)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["params"] = json.loads(str(d["params"]))
    d["scores"] = [d.pop(f"scores_{i:02d}") for i in range(int(d.pop("n_scores")))]
    return d


def golden_names(prefix=""):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def assert_scores_close(got, ref, rtol=SCORE_RTOL, what=""):
    got = np.asarray(got, dtype=np.float64).reshape(np.asarray(ref).shape)
    ref = np.asarray(ref, dtype=np.float64)
    scale = np.maximum(np.abs(ref), np.abs(ref).max() * 1e-6 + 1e-300)
    err = np.abs(got - ref) / scale
    assert np.all(np.isfinite(got) == np.isfinite(ref)), f"{what}: finiteness differs"
    m = np.isfinite(ref)
    record_margin("score_rel_err", err[m].max(initial=0.0))
    assert err[m].max(initial=0.0) <= rtol, f"{what}: score rel err {err[m].max():.3e} > {rtol}"


def assert_argmax_tie_aware(got_idx, ref_scores, tie_rtol=TIE_RTOL, what=""):
    """got_idx[j] must be the oracle's argmax over axis 0, or a near-tie by the ORACLE's own scores."""
    ref_scores = np.asarray(ref_scores, dtype=np.float64)
    if ref_scores.ndim == 1:
        ref_scores = ref_scores[:, None]
    got_idx = np.asarray(got_idx).reshape(-1)
    ref_idx = np.argmax(ref_scores, axis=0)
    for j, (gi, ri) in enumerate(zip(got_idx, ref_idx)):
        if gi == ri:
            continue
        best, mine = ref_scores[ri, j], ref_scores[gi, j]
        gap = abs(best - mine) / max(abs(best), 1e-300)
        record_margin("tie_gap", gap)
        assert gap <= tie_rtol, f"{what}: block {j}: picked {gi}, oracle {ri}, oracle score gap {gap:.3e} > {tie_rtol}"
    record_margin(None, None, {"selections": int(got_idx.size), "differing_selections": int((got_idx != ref_idx).sum())})
    return int((got_idx != ref_idx).sum())


def assert_interval_parity(got, ref, cand_step_rel, what=""):
    """Intervals equal bit-for-bit, or (near-tie) off by at most one candidate grid step."""
    got = np.asarray(got, dtype=np.float64).reshape(-1)
    ref = np.asarray(ref, dtype=np.float64).reshape(-1)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)
    assert rel.max(initial=0.0) <= cand_step_rel, f"{what}: interval rel diff {rel.max():.3e}"
    return int((got != ref).sum())


def candidate_grid(eq_alpha, eq_beta, eq_n):
    """The fp32 multiplier table of the reference (linear.py:544)."""
    return np.array([eq_alpha + i * (eq_beta - eq_alpha) / eq_n for i in range(eq_n + 1)], dtype=np.float32)


# Two intervals taken from the SAME fp32 candidate table differ by an exact ratio mult[a] / mult[b] up to the rounding of the
# two products (GRID_TOL).  Where the initial min-max interval itself comes from tensors captured on different hardware (this
# GPU's fp32 GEMMs vs the reference's CPU run) it can differ in its last bits, and with it every entry of the table: CAPTURE_TOL.
GRID_TOL = 4e-7
CAPTURE_TOL = 1.2e-6
MAX_GRID_STEPS = 1       # a differing interval lies at most this many entries of the candidate table from the reference's


def assert_on_candidate_grid(got, ref, mult, what="", tol=GRID_TOL, max_steps=MAX_GRID_STEPS, ref_scores=None, tie_rtol=TIE_RTOL):
    """Every interval is bit-identical to the reference's, or -- where the search settled on a different (near-tied)
    candidate -- it is another entry of the SAME candidate table: got / ref = mult[a] / mult[b] for some a, b (both are
    mult[.] * initial interval in fp32) with |a - b| <= max_steps: a near-tie sits next to the maximum of a smooth score curve.
    Further away only where the ORACLE's own table says so: `ref_scores` [candidates][blocks] is the reference's score table of
    the pass that selected `ref`; the entry `got` corresponds to must be within `tie_rtol` (relative) of that table's maximum --
    a flat optimum (the cosine metric on a handful of samples ties over many entries).  `max_steps=None`: no bound (callers
    comparing searches whose INPUTS differ -- tensors captured on other hardware -- say so at the call).
    Returns the number of blocks that differ; nothing else is tolerated."""
    got = np.asarray(got, dtype=np.float64).reshape(-1)
    ref = np.asarray(ref, dtype=np.float64).reshape(-1)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    m = np.asarray(mult, dtype=np.float64)[:-1]
    ratios = m[:, None] / m[None, :]
    dist = np.abs(np.arange(m.size)[:, None] - np.arange(m.size)[None, :])
    differ = 0
    for j, (g, r) in enumerate(zip(got, ref)):
        if g == r:
            continue
        differ += 1
        rel = np.abs(ratios - g / r) / (g / r)
        assert rel.min() <= tol, f"{what}: block {j}: interval {g!r} vs reference {r!r} is not on the candidate grid (off by {rel.min():.2e})"
        steps = int(dist[rel <= tol].min())
        record_margin("grid_steps", steps)
        if max_steps is not None and steps > max_steps and ref_scores is not None:
            tab = np.asarray(ref_scores, dtype=np.float64)
            tab = tab.reshape(tab.shape[0], -1)[: m.size]
            col = tab[:, j if tab.shape[1] > 1 else 0]
            ri = int(np.argmax(col))
            mine = int(np.argmin(np.abs(m / m[ri] - g / r)))
            gap = abs(col[ri] - col[mine]) / max(abs(col[ri]), 1e-300)
            record_margin("tie_gap_far", gap)
            assert gap <= tie_rtol, (f"{what}: block {j}: interval {g!r} vs reference {r!r}: {steps} entries of the candidate table apart and not "
                                     f"a tie by the oracle's own scores (candidate {mine} vs {ri}: gap {gap:.2e} > {tie_rtol})")
            continue
        assert max_steps is None or steps <= max_steps, (f"{what}: block {j}: interval {g!r} vs reference {r!r}: {steps} entries of the candidate "
                                                         f"table apart (bound {max_steps})")
    record_margin(None, None, {"intervals": int(got.size), "differing_intervals": differ})
    return differ


def grid_steps_between(got, want, mult, tol=GRID_TOL):
    """Smallest |a - b| over candidate pairs with mult[a] / mult[b] == got / want (to fp32 rounding): how many steps of the
    candidate table separate two intervals that were both taken from it.  None if no pair matches."""
    m = np.asarray(mult, dtype=np.float64)[:-1]
    r = float(got) / float(want)
    ratios = m[:, None] / m[None, :]
    hit = np.argwhere(np.abs(ratios - r) <= tol * r)
    return None if hit.size == 0 else int(np.abs(hit[:, 0] - hit[:, 1]).min())
