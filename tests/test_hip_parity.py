"""GPU parity: the HIP calibration engine (through the C ABI) vs the golden vectors of the reference
and vs the numpy oracle on seeded inputs.  Run on the MI355X box with `pytest -m gpu`.

Bar (floating point path): score tables within SCORE_RTOL of the reference's; the selected candidate equal
to the reference's argmax or a near-tie by the reference's own scores; intervals then
bit-identical to the reference's, or one candidate-grid step away at a near-tie.
"""
import numpy as np
import pytest
import torch

from tests.helpers import (assert_argmax_tie_aware, assert_on_candidate_grid, assert_scores_close, candidate_grid,
                           golden_names, load_golden)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from ptq4vit_amd import engine
    return engine


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _cmp_tables(pairs, name):
    """pairs: list of (got_scores[C, nblk], got_best[nblk], ref_table)."""
    flips = 0
    for i, (got, best, ref) in enumerate(pairs):
        ref2 = ref.reshape(ref.shape[0], -1)
        got = got[: ref2.shape[0], : ref2.shape[1]]
        assert_scores_close(got, ref2, what=f"{name}[{i}]")
        flips += assert_argmax_tie_aware(best[: ref2.shape[1]], ref2, what=f"{name}[{i}]")
        # the engine's own argmax must follow its own table (first max)
        np.testing.assert_array_equal(best[: ref2.shape[1]], np.argmax(got, axis=0), err_msg=f"{name}[{i}] select")
    return flips


def run_linear(eng, g, force_f32=False):
    p = dict(g["params"])
    p.pop("kind"); p.pop("oc")
    postgelu = p.pop("postgelu")
    w_iv, a_iv, scores, best = eng.linear_calibrate(
        weight=_t(g["weight"]), bias=_t(g["bias"]) if "bias" in g else None, x=_t(g["x"]), out=_t(g["out"]),
        grad=_t(g["grad"]), postgelu=postgelu, want_scores=True, force_f32=force_f32,
        n_H=p.pop("n_H", 1), n_a=p.pop("n_a", 1), **p)
    torch.cuda.synchronize()
    return w_iv.cpu().numpy(), a_iv.cpu().numpy(), scores.cpu().numpy(), best.cpu().numpy()


@pytest.mark.parametrize("force_f32", [False, True], ids=["i8", "f32"])
@pytest.mark.parametrize("name", golden_names("linear_") + golden_names("postgelu_"))
def test_linear_vs_reference_golden(eng, name, force_f32):
    g = load_golden(name)
    p = g["params"]
    nH, nA, R = p.get("n_H", 1), p.get("n_a", 1), p["search_round"]
    w_iv, a_iv, scores, best = run_linear(eng, g, force_f32)
    per_round = nH + nA
    pairs = []
    for r in range(R):
        pairs.append((scores[r, 0], best[r, 0], g["scores"][r * per_round + 0]))
        pairs.append((scores[r, 1][:, :1], best[r, 1][:1], g["scores"][r * per_round + nH]))
    flips = _cmp_tables(pairs, name)
    mult = candidate_grid(p["eq_alpha"], p["eq_beta"], p["eq_n"])
    # (further than one entry apart only where the reference's own last table shows a tie: helpers.assert_on_candidate_grid)
    last = (R - 1) * per_round
    moved = (assert_on_candidate_grid(w_iv, g["w_interval"], mult, name + " w_interval", ref_scores=g["scores"][last + nH - 1] if nH == 1 else None)
             + assert_on_candidate_grid(a_iv, g["a_interval"], mult, name + " a_interval", ref_scores=g["scores"][last + nH + nA - 1] if nA == 1 else None))
    print(f"[parity] {name} ({'f32' if force_f32 else 'i8'}): {flips} near-tie flips in the tables, {moved} intervals on another grid entry")
    if flips == 0 and nH == 1 and nA == 1:
        assert moved == 0        # same selections -> bit-identical intervals


@pytest.mark.parametrize("name", golden_names("matmul_"))
def test_matmul_vs_reference_golden(eng, name):
    g = load_golden(name)
    p = dict(g["params"])
    p.pop("kind")
    sos = p.pop("sos")
    R = p["search_round"]
    A_iv, B_iv, split, scores, best = eng.matmul_calibrate(
        A=_t(g["A"]), B=_t(g["B"]), out=_t(g["out"]), grad=_t(g["grad"]), sos=sos, want_scores=True, **p)
    torch.cuda.synchronize()
    scores, best = scores.cpu().numpy(), best.cpu().numpy()
    pairs = []
    for r in range(R):
        ta, tb = g["scores"][2 * r], g["scores"][2 * r + 1]
        if sos:
            pairs.append((scores[r, 0][:20, :1], best[r, 0][:1], ta))
        else:
            pairs.append((scores[r, 0], best[r, 0], ta))
        pairs.append((scores[r, 1], best[r, 1], tb))
    flips = _cmp_tables(pairs, name)
    if flips == 0:
        np.testing.assert_array_equal(B_iv.cpu().numpy(), g["B_interval"].reshape(-1))
        np.testing.assert_array_equal(A_iv.cpu().numpy(), np.asarray(g["A_interval"]).reshape(-1))
        if sos:
            assert float(split.cpu()) == float(g["split"])


@pytest.mark.parametrize("name", golden_names("conv_"))
def test_conv_vs_reference_golden(eng, name):
    g = load_golden(name)
    p = dict(g["params"])
    p.pop("kind")
    st = p.pop("stride")
    cw = p.pop("channelwise")
    R = p["search_round"]
    w_iv, a_iv, scores, best = eng.conv_calibrate(
        weight=_t(g["weight"]), bias=_t(g["bias"]), x=_t(g["x"]), out=_t(g["out"]), grad=_t(g["grad"]),
        stride=(st, st), padding=(0, 0), dilation=(1, 1), channelwise=cw, want_scores=True, **p)
    torch.cuda.synchronize()
    scores, best = scores.cpu().numpy(), best.cpu().numpy()
    aq = p["a_bit"] < 32
    per_round = 2 if aq else 1
    pairs = []
    for r in range(R):
        pairs.append((scores[r, 0], best[r, 0], g["scores"][r * per_round]))
        if aq:
            pairs.append((scores[r, 1][:, :1], best[r, 1][:1], g["scores"][r * per_round + 1]))
    flips = _cmp_tables(pairs, name)
    if flips == 0:
        np.testing.assert_array_equal(w_iv.cpu().numpy(), np.asarray(g["w_interval"]).reshape(-1))
        if aq:
            np.testing.assert_array_equal(a_iv.cpu().numpy(), np.asarray(g["a_interval"]).reshape(-1))


# ------------------------------------------------------------------------------------------------
# seeded multi-tile shapes vs the oracle (ragged M/N/K, several row/column tiles, n_V = 3)
# ------------------------------------------------------------------------------------------------
def _mk_linear(seed, b, T, K, N, postgelu=False, gscale=1e-3):
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((N, K)) * 0.05 * np.linspace(0.5, 2.0, N)[:, None]).astype(np.float32)
    bias = (rng.standard_normal(N) * 0.1).astype(np.float32)
    x = rng.standard_normal((b, T, K)).astype(np.float32)
    if postgelu:
        x = torch.nn.functional.gelu(torch.from_numpy(1.5 * x)).numpy()
    out = (x.reshape(-1, K) @ w.T + bias).reshape(b, T, N).astype(np.float32)
    grad = (rng.standard_normal(out.shape) * gscale).astype(np.float32)
    return w, bias, x, out, grad


@pytest.mark.parametrize("cfg", [
    dict(b=5, T=61, K=200, N=390, n_V=3, w_bit=8, a_bit=8, metric="hessian", postgelu=False),
    dict(b=5, T=61, K=200, N=384, n_V=3, w_bit=8, a_bit=8, metric="hessian", postgelu=False),   # fast sweep (32-aligned blocks)
    dict(b=4, T=70, K=330, N=200, n_V=1, w_bit=8, a_bit=8, metric="hessian", postgelu=True),    # fast sweep, twin, ragged N
    dict(b=4, T=70, K=330, N=256, n_V=2, w_bit=6, a_bit=6, metric="L1_norm", postgelu=False),
    dict(b=4, T=70, K=330, N=256, n_V=2, w_bit=8, a_bit=8, metric="linear_weighted_L2_norm", postgelu=False),
    dict(b=5, T=61, K=200, N=390, n_V=3, w_bit=6, a_bit=6, metric="hessian", postgelu=False),
    dict(b=4, T=70, K=330, N=130, n_V=1, w_bit=8, a_bit=8, metric="hessian", postgelu=True),
    dict(b=4, T=70, K=330, N=130, n_V=1, w_bit=6, a_bit=6, metric="hessian", postgelu=True),
    dict(b=5, T=61, K=200, N=390, n_V=3, w_bit=8, a_bit=8, metric="cosine", postgelu=False),
    dict(b=3, T=50, K=192, N=192, n_V=3, w_bit=8, a_bit=8, metric="cosine", postgelu=False),    # cosine on k_sweep6 (plain + transposed)
    dict(b=2, T=70, K=768, N=200, n_V=1, w_bit=6, a_bit=6, metric="cosine", postgelu=False),    # ... KT = 12, ragged N
    dict(b=3, T=50, K=96, N=160, n_V=2, w_bit=8, a_bit=8, metric="L2_norm", postgelu=False),
    # K = 192 / 384 / 768 bytes: register-stationary sweep (k_sweep6<KT = 3 / 6 / 12>), every epilogue flavour
    dict(b=3, T=50, K=192, N=128, n_V=2, w_bit=8, a_bit=8, metric="L1_norm", postgelu=False),
    dict(b=3, T=50, K=192, N=130, n_V=1, w_bit=8, a_bit=8, metric="L2_norm", postgelu=False),
    dict(b=3, T=50, K=384, N=192, n_V=3, w_bit=8, a_bit=8, metric="linear_weighted_L2_norm", postgelu=False),
    dict(b=3, T=50, K=384, N=200, n_V=1, w_bit=6, a_bit=6, metric="hessian", postgelu=False),
    dict(b=2, T=70, K=768, N=300, n_V=1, w_bit=8, a_bit=8, metric="square_weighted_L2_norm", postgelu=False),
    dict(b=2, T=70, K=768, N=192, n_V=3, w_bit=8, a_bit=8, metric="hessian", postgelu=False),
    dict(b=3, T=49, K=256, N=128, n_V=1, w_bit=8, a_bit=8, metric="hessian", postgelu=False),        # KT = 4 (Swin)
    dict(b=3, T=49, K=512, N=192, n_V=3, w_bit=8, a_bit=8, metric="L1_norm", postgelu=False),        # KT = 8 (Swin)
    dict(b=3, T=50, K=384, N=200, n_V=1, w_bit=8, a_bit=8, metric="hessian", postgelu=False, eq_n=3),     # k_sweep6, tiny candidate sets
    dict(b=3, T=50, K=768, N=192, n_V=3, w_bit=8, a_bit=8, metric="L2_norm", postgelu=False, eq_n=37),
    dict(b=3, T=50, K=192, N=96, n_V=1, w_bit=8, a_bit=8, metric="hessian", postgelu=False, eq_n=1),
    # K >= 1024: weight search on k_sweep2g (two candidates per pass, epilogue operands streamed), twin and plain,
    # odd candidate count (last pair runs its candidate twice)
    dict(b=2, T=70, K=1024, N=200, n_V=1, w_bit=8, a_bit=8, metric="hessian", postgelu=True),
    dict(b=2, T=70, K=1024, N=192, n_V=3, w_bit=8, a_bit=8, metric="L1_norm", postgelu=False),
    dict(b=2, T=70, K=1088, N=130, n_V=1, w_bit=6, a_bit=6, metric="hessian", postgelu=True, eq_n=37),
    dict(b=2, T=70, K=1024, N=256, n_V=2, w_bit=8, a_bit=8, metric="linear_weighted_L2_norm", postgelu=False, eq_n=51),
    dict(b=2, T=197, K=3072, N=768, n_V=1, w_bit=8, a_bit=8, metric="hessian", postgelu=True),          # ViT-B fc2 at 2 images
    # k_sweep7 (K >= 1024, K % 256 == 0, N % 32 == 0): ragged feature tiles (N = 320: 1.25 tiles of 256) and sample tiles,
    # every epilogue flavour, plain and twin, V blocks, tiny / odd candidate sets
    dict(b=3, T=131, K=2048, N=320, n_V=1, w_bit=8, a_bit=8, metric="hessian", postgelu=False),
    dict(b=3, T=131, K=1024, N=320, n_V=5, w_bit=8, a_bit=8, metric="L2_norm", postgelu=False),
    dict(b=3, T=131, K=1024, N=288, n_V=3, w_bit=6, a_bit=6, metric="L1_norm", postgelu=True),
    dict(b=2, T=150, K=1536, N=384, n_V=1, w_bit=8, a_bit=8, metric="linear_weighted_L2_norm", postgelu=True),   # ViT-S fc2 geometry
    dict(b=2, T=150, K=1024, N=256, n_V=2, w_bit=8, a_bit=8, metric="square_weighted_L2_norm", postgelu=False),
    dict(b=2, T=150, K=1024, N=64, n_V=1, w_bit=8, a_bit=8, metric="hessian", postgelu=True, eq_n=37),
    dict(b=1, T=300, K=4096, N=96, n_V=3, w_bit=8, a_bit=8, metric="hessian", postgelu=False, eq_n=1),
    dict(b=1, T=257, K=1024, N=1024, n_V=1, w_bit=8, a_bit=8, metric="hessian", postgelu=False, eq_n=9),       # 4 feature tiles
], ids=lambda c: f"{c['metric']}-w{c['w_bit']}-{'gelu' if c['postgelu'] else 'plain'}-K{c['K']}-N{c['N']}-nV{c['n_V']}")
def test_linear_multitile_vs_oracle(eng, cfg):
    from oracle.ptq4vit_oracle import LinearOracle
    cfg = dict(cfg)
    b, T, K, N, postgelu = (cfg.pop(k) for k in ("b", "T", "K", "N", "postgelu"))
    w, bias, x, out, grad = _mk_linear(7, b, T, K, N, postgelu)
    hp = dict(eq_alpha=0.01, eq_beta=1.2, eq_n=cfg.pop("eq_n", 100), search_round=2)
    o = LinearOracle(w, bias, postgelu=postgelu, **cfg, **hp)
    o.calibration_step2(x, out, grad)
    w_iv, a_iv, scores, best = eng.linear_calibrate(weight=_t(w), bias=_t(bias), x=_t(x), out=_t(out), grad=_t(grad),
                                                    postgelu=postgelu, n_H=1, n_a=1, want_scores=True, **cfg, **hp)
    torch.cuda.synchronize()
    scores, best = scores.cpu().numpy(), best.cpu().numpy()
    pairs = []
    for r in range(2):
        pairs.append((scores[r, 0], best[r, 0], o.trace[2 * r][1]))
        pairs.append((scores[r, 1][:, :1], best[r, 1][:1], o.trace[2 * r + 1][1]))
    flips = _cmp_tables(pairs, "multitile")
    if flips == 0:
        np.testing.assert_array_equal(w_iv.cpu().numpy(), o.w_interval.reshape(-1))
        np.testing.assert_array_equal(a_iv.cpu().numpy(), o.a_interval.reshape(-1))


@pytest.mark.parametrize("K,N,nV,bit,seed", [(384, 128, 2, 6, 2), (384, 128, 2, 6, 3), (768, 64, 2, 4, 3), (768, 96, 3, 8, 0)])
def test_postgelu_activation_search_follows_the_weight_interval_between_rounds(eng, K, N, nV, bit, seed):
    """Post-GELU twin on the register-stationary sweep (K = 384 / 768 B), three rounds, with inputs on which the weight
    interval MOVES between rounds 1 and 2 (asserted on the oracle's trace): the folded target of the twin activation
    search (raw_out - bias - s_neg s_w x_neg.W_q) depends on the current w_interval, so nothing derived from it may be kept
    across rounds (round 2 kept k_sweep6's epilogue image of round 1's target).  Tables of all six passes, and the
    memoised fused call without tables, against the oracle."""
    from oracle.ptq4vit_oracle import LinearOracle
    w, bias, x, out, grad = _mk_linear(seed, 2, 40, K, N, postgelu=True)
    hp = dict(w_bit=bit, a_bit=bit, metric="hessian", n_V=nV, eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=3)
    o = LinearOracle(w, bias, postgelu=True, **hp)
    o.calibration_step2(x, out, grad)
    w_sel = [np.argmax(o.trace[2 * r][1].reshape(100, -1), axis=0) for r in range(3)]
    assert not np.array_equal(w_sel[0], w_sel[1]), "test input: the weight interval must move between the rounds"
    args = dict(weight=_t(w), bias=_t(bias), x=_t(x), out=_t(out), grad=_t(grad), postgelu=True, n_H=1, n_a=1)
    w_iv, a_iv, scores, best = eng.linear_calibrate(want_scores=True, **args, **hp)
    w2, a2 = eng.linear_calibrate(want_scores=False, **args, **hp)[:2]
    torch.cuda.synchronize()
    scores, best = scores.cpu().numpy(), best.cpu().numpy()
    pairs = []
    for r in range(3):
        pairs.append((scores[r, 0], best[r, 0], o.trace[2 * r][1]))
        pairs.append((scores[r, 1][:, :1], best[r, 1][:1], o.trace[2 * r + 1][1]))
    flips = _cmp_tables(pairs, "postgelu-rounds")
    assert torch.equal(w_iv, w2) and torch.equal(a_iv, a2), "memoised call differs from the call with score tables"
    if flips == 0:
        np.testing.assert_array_equal(w_iv.cpu().numpy(), o.w_interval.reshape(-1))
        np.testing.assert_array_equal(a_iv.cpu().numpy(), o.a_interval.reshape(-1))


def test_quantize_i8_bit_exact(eng):
    """Integer planes are bit-exact: clamp(rint(x/s)) with IEEE division and round-half-even."""
    from oracle.ptq4vit_oracle import quant_int
    rng = np.random.default_rng(0)
    x = rng.standard_normal((257, 199)).astype(np.float32) * 3
    # plant exact half-way cases: x = (k + 0.5) * s
    s = np.array([0.0371, 0.011, 0.5], dtype=np.float32)
    x[0, :100] = (np.arange(100, dtype=np.float32) - 50 + 0.5) * s[0]
    q = eng.quantize_i8(_t(x), _t(s), 100, -128, 127).cpu().numpy()
    ref = quant_int(x, np.repeat(s, 100)[:257, None], -128, 127)
    np.testing.assert_array_equal(q, ref)
    q6 = eng.quantize_i8(_t(x), _t(s), 100, -32, 31).cpu().numpy()
    np.testing.assert_array_equal(q6, quant_int(x, np.repeat(s, 100)[:257, None], -32, 31))


def test_determinism_and_f32_cross_check(eng):
    """Run-to-run bit-identical results; the int8 MFMA path and the fp32 MFMA path pick the same candidates."""
    w, bias, x, out, grad = _mk_linear(11, 8, 197, 192, 576, gscale=1e-10)
    hp = dict(w_bit=8, a_bit=8, metric="hessian", eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=3, n_V=3, n_H=1, n_a=1)
    args = dict(weight=_t(w), bias=_t(bias), x=_t(x), out=_t(out), grad=_t(grad), want_scores=True)
    r1 = eng.linear_calibrate(**args, **hp)
    r2 = eng.linear_calibrate(**args, **hp)
    r3 = eng.linear_calibrate(**args, **hp, force_f32=True)
    torch.cuda.synchronize()
    for a, b in zip(r1, r2):
        assert torch.equal(a, b)
    s1, s3 = r1[2].cpu().numpy(), r3[2].cpu().numpy()
    assert_scores_close(s1, s3, rtol=5e-4, what="i8 vs f32 path")
    for r in range(3):
        assert_argmax_tie_aware(r1[3][r, 0].cpu().numpy(), s3[r, 0], what="w")
        assert_argmax_tie_aware(r1[3][r, 1][:1].cpu().numpy(), s3[r, 1][:, :1], what="a")


def test_fast_sweep_matches_generic_sweep(eng):
    """A/B: k_sweep2 (LDS-DMA ring, one float per wave) vs the generic k_sweep on the same launch sequence."""
    w, bias, x, out, grad = _mk_linear(13, 6, 197, 384, 768, gscale=1e-3)
    hp = dict(w_bit=8, a_bit=8, metric="hessian", eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=2, n_V=3, n_H=1, n_a=1)
    args = dict(weight=_t(w), bias=_t(bias), x=_t(x), out=_t(out), grad=_t(grad), want_scores=True)
    fast = eng.linear_calibrate(**args, **hp)
    eng.debug_variant(0, force_generic=True)
    try:
        slow = eng.linear_calibrate(**args, **hp)
    finally:
        eng.debug_variant(0)
    torch.cuda.synchronize()
    assert_scores_close(fast[2].cpu().numpy(), slow[2].cpu().numpy(), rtol=1e-5, what="fast vs generic")
    assert torch.equal(fast[3], slow[3])
    assert torch.equal(fast[0], slow[0]) and torch.equal(fast[1], slow[1])


# ------------------------------------------------------------------------------------------------
# MatMul / split-of-softmax / Conv at the BASELINE shapes (several ragged 128-tiles per operand), against the oracle
# ------------------------------------------------------------------------------------------------
def _mk_attention(seed, b, H, S, D, kind, gscale=1e-3):
    rng = np.random.default_rng(seed)
    if kind == "qk":      # q . k^T (reference utils/models.py:16-18): A (b,H,S,D), B = k.transpose view (b,H,D,S)
        A = rng.standard_normal((b, H, S, D)).astype(np.float32) * (D ** -0.5) * np.linspace(0.5, 2.0, H, dtype=np.float32)[None, :, None, None]
        kmat = rng.standard_normal((b, H, S, D)).astype(np.float32) * np.linspace(2.0, 0.7, H, dtype=np.float32)[None, :, None, None]
        B = kmat.transpose(0, 1, 3, 2)
    else:                 # softmax(scores) . v: A (b,H,S,S) rows sum to 1, B (b,H,S,D)
        A = torch.softmax(torch.from_numpy(rng.standard_normal((b, H, S, S)).astype(np.float32) * 3), -1).numpy()
        B = rng.standard_normal((b, H, S, D)).astype(np.float32) * np.linspace(0.5, 2.0, H, dtype=np.float32)[None, :, None, None]
    out = (A @ B).astype(np.float32)
    grad = (rng.standard_normal(out.shape) * gscale).astype(np.float32)
    return A, B, out, grad


@pytest.mark.parametrize("cfg", [
    dict(kind="qk", b=2, H=12, S=197, D=64, bit=8, metric="hessian"),           # ViT-B matmul1: 197x64x197, Z = 24 (2 x 2 ragged tiles)
    dict(kind="qk", b=2, H=12, S=197, D=64, bit=6, metric="hessian"),           # ... W6A6
    dict(kind="sv", b=2, H=12, S=197, D=64, bit=8, metric="hessian"),           # ViT-B matmul2 (SoS): 197x197x64 + the 20-split fp32 search
    dict(kind="sv", b=2, H=12, S=197, D=64, bit=6, metric="hessian"),
    dict(kind="qk", b=64, H=4, S=144, D=32, bit=8, metric="hessian"),           # Swin-B stage 1: one image = 64 windows, 144x32x144
    dict(kind="sv", b=64, H=4, S=144, D=32, bit=8, metric="hessian"),           # ... its split-of-softmax matmul
    dict(kind="qk", b=1, H=6, S=197, D=64, bit=8, metric="cosine"),             # BasePTQ metric, ViT-S heads
    dict(kind="sv", b=1, H=3, S=577, D=64, bit=8, metric="hessian"),            # 384-resolution token count (577 = 4.5 tiles)
    dict(kind="qk", b=2, H=2, S=257, D=64, bit=8, metric="hessian"),            # 257..304 tokens: 17x17 blocks would fit k_sweep9's block budget but not its 256-row stage
    dict(kind="qk", b=1, H=3, S=300, D=32, bit=8, metric="hessian"),
    dict(kind="qk", b=1, H=2, S=577, D=64, bit=8, metric="hessian"),            # ViT-B/384 q.k^T
], ids=lambda c: f"{c['kind']}-{c['metric']}-{c['bit']}bit-b{c['b']}H{c['H']}S{c['S']}D{c['D']}")
def test_matmul_baseline_shapes_vs_oracle(eng, cfg):
    """Same bar as the Linear size tests: every score table of every pass within SCORE_RTOL of the oracle's, selections
    equal or near-ties by the oracle's own scores, intervals then bit-identical (reference matmul.py:483-563, 600-631)."""
    from oracle.ptq4vit_oracle import MatMulOracle
    sos = cfg["kind"] == "sv"
    A, B, out, grad = _mk_attention(5, cfg["b"], cfg["H"], cfg["S"], cfg["D"], cfg["kind"])
    hp = dict(A_bit=cfg["bit"], B_bit=cfg["bit"], metric=cfg["metric"], eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=2)
    o = MatMulOracle(sos=sos, **hp)
    res = o.calibration_step2(A, B, out, grad)
    Bt = _t(np.ascontiguousarray(B.transpose(0, 1, 3, 2))).transpose(-2, -1) if cfg["kind"] == "qk" else _t(B)   # k.transpose VIEW
    A_iv, B_iv, split, scores, best = eng.matmul_calibrate(A=_t(A), B=Bt, out=_t(out), grad=_t(grad), sos=sos,
                                                           want_scores=True, **hp)
    torch.cuda.synchronize()
    scores, best = scores.cpu().numpy(), best.cpu().numpy()
    pairs = []
    for r in range(2):
        ta, tb = o.trace[2 * r][1], o.trace[2 * r + 1][1]
        pairs.append((scores[r, 0][:20, :1], best[r, 0][:1], ta) if sos else (scores[r, 0], best[r, 0], ta))
        pairs.append((scores[r, 1], best[r, 1], tb))
    flips = _cmp_tables(pairs, "matmul-baseline")
    print(f"[parity] matmul {cfg}: {flips} near-tie flips")
    if flips == 0:
        np.testing.assert_array_equal(B_iv.cpu().numpy(), np.asarray(res["B_interval"]).reshape(-1))
        np.testing.assert_array_equal(A_iv.cpu().numpy(), np.asarray(res["A_interval"]).reshape(-1))
        if sos:
            assert float(split.cpu()) == float(res["split"])


@pytest.mark.parametrize("cfg", [
    dict(b=4, hw=224, oc=768, k=16, channelwise=True, w_bit=8, metric="hessian"),     # ViT-B patch embedding: (eq_n, 768) table, M = 784
    dict(b=4, hw=224, oc=768, k=16, channelwise=True, w_bit=6, metric="hessian"),
    dict(b=3, hw=224, oc=384, k=16, channelwise=True, w_bit=8, metric="cosine"),      # channel-wise cosine over the pixels of an image
    dict(b=4, hw=224, oc=192, k=16, channelwise=False, w_bit=8, metric="cosine"),     # BasePTQ / DeiT-tiny: layer-wise, cosine over oc
    dict(b=4, hw=224, oc=192, k=16, channelwise=False, w_bit=8, metric="hessian"),
    dict(b=2, hw=384, oc=128, k=4, channelwise=True, w_bit=8, metric="hessian"),      # Swin-B/384 patch embedding: 4x4 patches, K = 48, M = 18432
], ids=lambda c: f"{'cw' if c['channelwise'] else 'lw'}-{c['metric']}-w{c['w_bit']}-b{c['b']}-{c['hw']}-oc{c['oc']}-k{c['k']}")
def test_conv_baseline_shapes_vs_oracle(eng, cfg):
    """Patch-embedding searches at full image size (reference conv.py:526-557 channel-wise, 365-396 layer-wise): the
    whole (eq_n, oc) score table against the oracle."""
    from oracle.ptq4vit_oracle import ConvOracle
    rng = np.random.default_rng(9)
    b, hw, oc, k = cfg["b"], cfg["hw"], cfg["oc"], cfg["k"]
    w = (rng.standard_normal((oc, 3, k, k)) * 0.05 * np.linspace(0.3, 3.0, oc)[:, None, None, None]).astype(np.float32)
    bias = (rng.standard_normal(oc) * 0.1).astype(np.float32)
    x = rng.standard_normal((b, 3, hw, hw)).astype(np.float32)
    out = torch.nn.functional.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(bias), stride=k).numpy()
    grad = (rng.standard_normal(out.shape) * 1e-3).astype(np.float32)
    hp = dict(w_bit=cfg["w_bit"], a_bit=32, metric=cfg["metric"], eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=1)
    o = ConvOracle(w, bias, stride=k, channelwise=cfg["channelwise"], **hp)
    res = o.calibration_step2(x, out, grad)
    w_iv, a_iv, scores, best = eng.conv_calibrate(weight=_t(w), bias=_t(bias), x=_t(x), out=_t(out), grad=_t(grad),
                                                  stride=(k, k), padding=(0, 0), dilation=(1, 1),
                                                  channelwise=cfg["channelwise"], want_scores=True, **hp)
    torch.cuda.synchronize()
    flips = _cmp_tables([(scores[0, 0].cpu().numpy(), best[0, 0].cpu().numpy(), o.trace[0][1])], "conv-baseline")
    print(f"[parity] conv {cfg}: {flips} near-tie flips of {scores.shape[-1]} columns")
    got, ref = w_iv.cpu().numpy(), np.asarray(res["w_interval"]).reshape(-1)
    moved = assert_on_candidate_grid(got, ref, candidate_grid(0.01, 1.2, 100), "conv w_interval", ref_scores=o.trace[0][1])
    assert moved <= flips


def test_pass_memo_is_exact_with_column_blocks_and_activation_groups(eng):
    """n_H = 2 / n_a = 2, three rounds: the searches are coordinate descents whose result depends on the interval
    ENTERING the pass (linear.py:468), so they must not be restored from the memo -- memo on == memo off, bit for bit."""
    w, bias, x, out, grad = _mk_linear(21, 4, 50, 128, 96)
    hp = dict(w_bit=8, a_bit=8, metric="hessian", eq_alpha=0.01, eq_beta=1.2, eq_n=40, search_round=3, n_V=2, n_H=2, n_a=2)
    args = dict(weight=_t(w), bias=_t(bias), x=_t(x), out=_t(out), grad=_t(grad))
    on = eng.linear_calibrate(**args, **hp, memoize=True)
    off = eng.linear_calibrate(**args, **hp, memoize=False)
    torch.cuda.synchronize()
    assert torch.equal(on[0], off[0]) and torch.equal(on[1], off[1])


def test_large_k_sweep_matches_the_128_tile_sweeps(eng):
    """A/B at ViT-B fc2 geometry (8 images): k_sweep7 (256 x 256 tiles, streamed epilogue operands) against k_sweep2 /
    k_sweep2g (variant 32768) -- same selections, score tables equal to fp32 summation-order noise."""
    w, bias, x, out, grad = _mk_linear(17, 8, 197, 3072, 768, postgelu=True, gscale=1e-3)
    hp = dict(w_bit=8, a_bit=8, metric="hessian", eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=2, n_V=1, n_H=1, n_a=1)
    args = dict(weight=_t(w), bias=_t(bias), x=_t(x), out=_t(out), grad=_t(grad), want_scores=True, postgelu=True)
    new = eng.linear_calibrate(**args, **hp)
    again = eng.linear_calibrate(**args, **hp)
    eng.debug_variant(32768)
    try:
        old = eng.linear_calibrate(**args, **hp)
    finally:
        eng.debug_variant(0)
    torch.cuda.synchronize()
    for a, b in zip(new, again):
        assert torch.equal(a, b), "k_sweep7 is not run-to-run deterministic"
    assert_scores_close(new[2].cpu().numpy(), old[2].cpu().numpy(), rtol=2e-5, what="k_sweep7 vs k_sweep2")
    assert torch.equal(new[3], old[3]) and torch.equal(new[0], old[0]) and torch.equal(new[1], old[1])


@pytest.mark.parametrize("b,T,K,N,bit", [(8, 197, 3072, 768, 8), (3, 131, 1024, 288, 6), (2, 150, 1536, 384, 8), (1, 100, 4096, 64, 4)],
                         ids=["vit-b-fc2", "ragged-w6", "vit-s-fc2", "one-sample-tile-w4"])
def test_merged_plane_twin_is_bit_identical_to_the_two_plane_twin(eng, b, T, K, N, bit):
    """A/B of the post-GELU weight search on k_sweep7: ONE merged int8 plane k_pos + k_neg split into its two fragments in
    registers (default) against the two streamed planes (variant 2097152).  The int32 accumulators are the same integers and
    the epilogue is the same code in the same order: score tables, selections and intervals must be BIT-identical."""
    w, bias, x, out, grad = _mk_linear(23, b, T, K, N, postgelu=True, gscale=1e-3)
    hp = dict(w_bit=bit, a_bit=bit, metric="hessian", eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=2, n_V=1, n_H=1, n_a=1)
    args = dict(weight=_t(w), bias=_t(bias), x=_t(x), out=_t(out), grad=_t(grad), want_scores=True, postgelu=True)
    new = eng.linear_calibrate(**args, **hp)
    again = eng.linear_calibrate(**args, **hp)
    eng.debug_variant(2097152)
    try:
        old = eng.linear_calibrate(**args, **hp)
    finally:
        eng.debug_variant(0)
    torch.cuda.synchronize()
    for a, c in zip(new, again):
        assert torch.equal(a, c), "merged-plane twin is not run-to-run deterministic"
    for a, c, what in zip(new, old, ("w_interval", "a_interval", "score tables", "selections")):
        assert torch.equal(a, c), f"merged-plane twin differs from the two-plane twin: {what}"


def test_single_ktile_sweep_matches_the_streaming_sweep(eng):
    """A/B at ViT-B q.k^T geometry (4 images x 12 heads, 197 x 64 x 197): k_sweep8 (fixed operand in registers, 8-deep ring)
    against k_sweep2 (variant 65536), both searches -- identical selections, tables equal to summation-order noise (the
    epilogue arithmetic is the same code)."""
    A, B, out, grad = _mk_attention(23, 4, 12, 197, 64, "qk")
    hp = dict(A_bit=8, B_bit=8, metric="hessian", eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=2)
    Bt = _t(np.ascontiguousarray(B.transpose(0, 1, 3, 2))).transpose(-2, -1)
    args = dict(A=_t(A), B=Bt, out=_t(out), grad=_t(grad), want_scores=True)
    new = eng.matmul_calibrate(**args, **hp)
    again = eng.matmul_calibrate(**args, **hp)
    eng.debug_variant(65536)
    try:
        old = eng.matmul_calibrate(**args, **hp)
    finally:
        eng.debug_variant(0)
    torch.cuda.synchronize()
    for a, b in zip(new, again):
        if a is not None:
            assert torch.equal(a, b), "k_sweep8 is not run-to-run deterministic"
    assert_scores_close(new[3].cpu().numpy(), old[3].cpu().numpy(), rtol=1e-6, what="k_sweep8 vs k_sweep2")
    assert torch.equal(new[4], old[4]) and torch.equal(new[0], old[0]) and torch.equal(new[1], old[1])


@pytest.mark.parametrize("shape,metric", [((4, 12, 197, 64), "hessian"), ((16, 4, 144, 32), "hessian"),
                                          ((8, 3, 49, 32), "L2_norm"), ((2, 12, 197, 64), "linear_weighted_L2_norm")],
                         ids=["vit-b", "swin-w12", "swin-w7-l2", "vit-b-linear-weighted"])
def test_split_search_kernel_matches_the_generic_fp32_sweep(eng, shape, metric):
    """A/B of the split-of-softmax split search (matmul.py:600-631): k_sos_split (A quantised in registers from two
    candidate-invariant images per element, one kernel) against 20 fp32 planes through k_pack + the generic fp32 sweep
    (variant 131072): same split, same A_interval, same B_interval afterwards, score tables equal to summation-order noise."""
    b, H, S, D = shape
    A, B, out, grad = _mk_attention(31, b, H, S, D, "sv")
    hp = dict(A_bit=8, B_bit=8, metric=metric, eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=1, sos=True)
    args = dict(A=_t(A), B=_t(B), out=_t(out), grad=_t(grad), want_scores=True)
    new = eng.matmul_calibrate(**args, **hp)
    again = eng.matmul_calibrate(**args, **hp)
    eng.debug_variant(131072)
    try:
        old = eng.matmul_calibrate(**args, **hp)
    finally:
        eng.debug_variant(0)
    torch.cuda.synchronize()
    for a, b_ in zip(new, again):
        if a is not None:
            assert torch.equal(a, b_), "k_sos_split is not run-to-run deterministic"
    assert torch.equal(new[2], old[2]) and torch.equal(new[0], old[0]), (new[2], old[2])       # split, A_interval
    sn, so = new[3].cpu().numpy(), old[3].cpu().numpy()       # (round, search, eq_n, H): the split table is [0, 0, :20, 0]
    assert_scores_close(sn[0, 0, :20, 0], so[0, 0, :20, 0], rtol=2e-5, what="k_sos_split vs fp32 sweep (split table)")
    assert_scores_close(sn[0, 1], so[0, 1], rtol=1e-6, what="B search after the split search")
    assert torch.equal(new[1], old[1]) and torch.equal(new[4], old[4])


@pytest.mark.parametrize("kind,b,H,S,D", [("qk", 64, 4, 144, 32), ("qk", 8, 12, 197, 64), ("sv", 64, 4, 144, 32), ("sv", 8, 12, 197, 64),
                                           ("qk", 96, 3, 49, 32)],
                         ids=["qk-swin-w12", "qk-vit-b", "sv-swin-w12", "sv-vit-b", "qk-swin-w7"])
def test_attention_sweeps_are_run_to_run_deterministic(eng, kind, b, H, S, D):
    """Six calibrations of the same attention matmul must agree bit for bit (score tables, selections, intervals).  Token counts
    that leave most of a 128 x 128 tile as padding make some workgroups step through their candidates much faster than the
    others: the first version of k_sweep8 proved the landing of the wrong ring stage and lost this test two runs out of three."""
    A, B, out, grad = _mk_attention(41, b, H, S, D, kind)
    hp = dict(A_bit=8, B_bit=8, metric="hessian", eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=2, sos=(kind == "sv"))
    Bt = _t(np.ascontiguousarray(B.transpose(0, 1, 3, 2))).transpose(-2, -1) if kind == "qk" else _t(B)
    args = dict(A=_t(A), B=Bt, out=_t(out), grad=_t(grad), want_scores=True)
    first = eng.matmul_calibrate(**args, **hp)
    for run in range(5):
        again = eng.matmul_calibrate(**args, **hp)
        torch.cuda.synchronize()
        for x, y in zip(first, again):
            if x is not None:
                assert torch.equal(x, y), f"run {run + 2} differs from run 1"


@pytest.mark.parametrize("b,T,K,N,nV,gelu", [(1, 197, 768, 768, 1, False), (2, 50, 3072, 768, 1, True), (1, 10, 384, 1152, 3, False),
                                             (3, 197, 1024, 1024, 1, False), (1, 5, 192, 192, 1, False)],
                         ids=["proj-1img", "fc2-twin-100rows", "qkv-s-10rows", "vit-l-proj", "tiny"])
def test_linear_sweeps_are_run_to_run_deterministic(eng, b, T, K, N, nV, gelu):
    """The same for the Linear sweeps at row counts that leave most tiles empty (k_sweep6 / k_sweep7 / k_sweep2g rings)."""
    w, bias, x, out, grad = _mk_linear(43, b, T, K, N, postgelu=gelu)
    hp = dict(w_bit=8, a_bit=8, metric="hessian", eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=2, n_V=nV, n_H=1, n_a=1)
    args = dict(weight=_t(w), bias=_t(bias), x=_t(x), out=_t(out), grad=_t(grad), postgelu=gelu, want_scores=True)
    first = eng.linear_calibrate(**args, **hp)
    for run in range(4):
        again = eng.linear_calibrate(**args, **hp)
        torch.cuda.synchronize()
        for a_, b_ in zip(first, again):
            if a_ is not None:
                assert torch.equal(a_, b_), f"run {run + 2} differs from run 1"


def _random_linear_cases(n=18, seed=2024):
    """Seeded random Linear geometries across every dispatch branch of the sweeps: K in the register-stationary set
    (192 / 256 / 384 / 512 / 768), other K <= 768 (LDS-stationary), K >= 1024 (multiples of 256 and not), ragged rows / features,
    1-3 row blocks, both bit widths, every difference metric, post-GELU twin or not, few candidates."""
    rng = np.random.default_rng(seed)
    ks = [192, 256, 384, 512, 768, 64, 100, 320, 640, 1024, 1280, 1536, 2048, 1100]
    metrics = ["hessian", "L2_norm", "L1_norm", "linear_weighted_L2_norm", "square_weighted_L2_norm"]
    cases = []
    for i in range(n):
        K = int(ks[i % len(ks)])
        nV = int(rng.choice([1, 1, 2, 3]))
        N = int(rng.integers(1, 9)) * 32 * nV if rng.random() < 0.6 else int(rng.integers(5, 200)) * nV
        cases.append(dict(b=int(rng.integers(1, 4)), T=int(rng.integers(3, 140)), K=K, N=N, n_V=nV,
                          bit=int(rng.choice([8, 8, 6, 4])), metric=str(rng.choice(metrics)), postgelu=bool(rng.random() < 0.35),
                          eq_n=int(rng.choice([100, 100, 37, 7])), seed=100 + i))
    return cases


@pytest.mark.parametrize("cfg", _random_linear_cases(), ids=lambda c: f"K{c['K']}-N{c['N']}-nV{c['n_V']}-T{c['b']}x{c['T']}-w{c['bit']}-{c['metric'][:6]}-{'gelu' if c['postgelu'] else 'plain'}-c{c['eq_n']}")
def test_random_linear_geometries_vs_oracle(eng, cfg):
    from oracle.ptq4vit_oracle import LinearOracle
    w, bias, x, out, grad = _mk_linear(cfg["seed"], cfg["b"], cfg["T"], cfg["K"], cfg["N"], cfg["postgelu"])
    hp = dict(eq_alpha=0.01, eq_beta=1.2, eq_n=cfg["eq_n"], search_round=2, w_bit=cfg["bit"], a_bit=cfg["bit"], n_V=cfg["n_V"],
              metric=cfg["metric"])
    o = LinearOracle(w, bias, postgelu=cfg["postgelu"], **hp)
    o.calibration_step2(x, out, grad)
    w_iv, a_iv, scores, best = eng.linear_calibrate(weight=_t(w), bias=_t(bias), x=_t(x), out=_t(out), grad=_t(grad),
                                                    postgelu=cfg["postgelu"], n_H=1, n_a=1, want_scores=True, **hp)
    torch.cuda.synchronize()
    scores, best = scores.cpu().numpy(), best.cpu().numpy()
    pairs = []
    for r in range(2):
        pairs.append((scores[r, 0], best[r, 0], o.trace[2 * r][1]))
        pairs.append((scores[r, 1][:, :1], best[r, 1][:1], o.trace[2 * r + 1][1]))
    flips = _cmp_tables(pairs, "random-linear")
    if flips == 0:
        np.testing.assert_array_equal(w_iv.cpu().numpy(), o.w_interval.reshape(-1))
        np.testing.assert_array_equal(a_iv.cpu().numpy(), o.a_interval.reshape(-1))


def _random_attention_cases(n=12, seed=77):
    """Seeded random attention geometries: head dims 16-96 (single k-tile sweep or the streaming one), token counts from 5 to
    300 (k_sos_split up to 200 keys, the generic fp32 split search above), 1-5 heads, both matmuls, 8 / 6 bit."""
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(n):
        cases.append(dict(kind="qk" if i % 2 == 0 else "sv", b=int(rng.integers(1, 4)), H=int(rng.integers(1, 6)),
                          S=int(rng.choice([5, 17, 49, 50, 64, 129, 144, 197, 201, 257, 300])), D=int(rng.choice([16, 32, 64, 96])),
                          bit=int(rng.choice([8, 8, 6])), metric=str(rng.choice(["hessian", "hessian", "L2_norm", "L1_norm"])), seed=300 + i))
    return cases


@pytest.mark.parametrize("cfg", _random_attention_cases(), ids=lambda c: f"{c['kind']}-b{c['b']}H{c['H']}S{c['S']}D{c['D']}-{c['bit']}bit-{c['metric'][:4]}")
def test_random_attention_geometries_vs_oracle(eng, cfg):
    from oracle.ptq4vit_oracle import MatMulOracle
    sos = cfg["kind"] == "sv"
    A, B, out, grad = _mk_attention(cfg["seed"], cfg["b"], cfg["H"], cfg["S"], cfg["D"], cfg["kind"])
    hp = dict(A_bit=cfg["bit"], B_bit=cfg["bit"], metric=cfg["metric"], eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=2)
    o = MatMulOracle(sos=sos, **hp)
    res = o.calibration_step2(A, B, out, grad)
    Bt = _t(np.ascontiguousarray(B.transpose(0, 1, 3, 2))).transpose(-2, -1) if cfg["kind"] == "qk" else _t(B)
    A_iv, B_iv, split, scores, best = eng.matmul_calibrate(A=_t(A), B=Bt, out=_t(out), grad=_t(grad), sos=sos, want_scores=True, **hp)
    torch.cuda.synchronize()
    scores, best = scores.cpu().numpy(), best.cpu().numpy()
    pairs = []
    for r in range(2):
        ta, tb = o.trace[2 * r][1], o.trace[2 * r + 1][1]
        pairs.append((scores[r, 0][:20, :1], best[r, 0][:1], ta) if sos else (scores[r, 0], best[r, 0], ta))
        pairs.append((scores[r, 1], best[r, 1], tb))
    flips = _cmp_tables(pairs, "random-attention")
    if flips == 0:
        np.testing.assert_array_equal(B_iv.cpu().numpy(), np.asarray(res["B_interval"]).reshape(-1))
        np.testing.assert_array_equal(A_iv.cpu().numpy(), np.asarray(res["A_interval"]).reshape(-1))
        if sos:
            assert float(split.cpu()) == float(res["split"])


@pytest.mark.parametrize("b,H,S,D", [(16, 4, 144, 32), (4, 12, 197, 64), (24, 3, 49, 32), (2, 2, 250, 64), (2, 2, 257, 64), (1, 3, 300, 32)],
                         ids=["swin-w12", "vit-b", "swin-w7", "250-tokens", "257-tokens-falls-back", "300-tokens-falls-back"])
def test_block16_sweep_matches_the_128_tile_sweep(eng, b, H, S, D):
    """A/B of the single-k-tile sweeps: k_sweep9 (16 x 16 blocks, forced with variant 1048576) against k_sweep8 (128 x 128
    tiles, variant 524288): same selections and intervals, score tables to summation-order noise, both searches."""
    A, B, out, grad = _mk_attention(57, b, H, S, D, "qk")
    hp = dict(A_bit=8, B_bit=8, metric="hessian", eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=2)
    Bt = _t(np.ascontiguousarray(B.transpose(0, 1, 3, 2))).transpose(-2, -1)
    args = dict(A=_t(A), B=Bt, out=_t(out), grad=_t(grad), want_scores=True)
    try:
        eng.debug_variant(1048576)
        new = eng.matmul_calibrate(**args, **hp)
        again = eng.matmul_calibrate(**args, **hp)
        eng.debug_variant(524288)
        old = eng.matmul_calibrate(**args, **hp)
    finally:
        eng.debug_variant(0)
    torch.cuda.synchronize()
    for x, y in zip(new, again):
        if x is not None:
            assert torch.equal(x, y), "k_sweep9 is not run-to-run deterministic"
    assert_scores_close(new[3].cpu().numpy(), old[3].cpu().numpy(), rtol=2e-6, what="k_sweep9 vs k_sweep8")
    assert torch.equal(new[4], old[4]) and torch.equal(new[0], old[0]) and torch.equal(new[1], old[1])


# ------------------------------------------------------------------------------------------------
# exact candidate pruning (p4v_api.hip::run_pass_pruned): same intervals with and without, bit for bit
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [
    dict(b=8, T=197, K=768, N=768, n_V=1, bit=8, metric="hessian", postgelu=False),          # ViT-B proj at 8 images: k_sweep6
    dict(b=8, T=197, K=768, N=2304, n_V=3, bit=8, metric="hessian", postgelu=False),         # qkv: three score blocks, three hulls
    dict(b=6, T=197, K=3072, N=768, n_V=1, bit=8, metric="hessian", postgelu=True),          # fc2: k_sweep7 plain + merged twin
    dict(b=6, T=197, K=768, N=3072, n_V=1, bit=6, metric="L2_norm", postgelu=False),         # fc1, W6A6, unweighted
    dict(b=5, T=131, K=1024, N=320, n_V=5, bit=8, metric="L1_norm", postgelu=False),         # ragged tiles, 5 blocks, |.| terms
    dict(b=16, T=50, K=384, N=200, n_V=1, bit=8, metric="linear_weighted_L2_norm", postgelu=True),   # twin on k_sweep6 sizes
    dict(b=3, T=40, K=192, N=96, n_V=1, bit=8, metric="hessian", postgelu=False),            # too few samples for a slice: unpruned
], ids=lambda c: f"K{c['K']}-N{c['N']}-nV{c['n_V']}-b{c['b']}-w{c['bit']}-{c['metric'][:6]}-{'gelu' if c['postgelu'] else 'plain'}")
def test_candidate_pruning_is_exact_linear(eng, cfg):
    """Stage A (all candidates, first eighth of the samples) + B1 (the bound) + B2 (the surviving range on all samples) must
    select exactly what the sweep of every candidate over every sample selects: three rounds, memo on, both searches."""
    cfg = dict(cfg)
    b, T, K, N, bit, postgelu = (cfg.pop(k) for k in ("b", "T", "K", "N", "bit", "postgelu"))
    w, bias, x, out, grad = _mk_linear(31, b, T, K, N, postgelu)
    hp = dict(w_bit=bit, a_bit=bit, eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=3, n_H=1, n_a=1, postgelu=postgelu, **cfg)
    # two weight profiles: spread evenly over the samples (loose bounds: many survivors -- the engine would not prune such a
    # module by itself, variant 8388608 forces it) and concentrated on a few samples as in a ViT (tight bounds)
    rng = np.random.default_rng(5)
    heavy = grad.copy().reshape(-1, N)
    heavy[rng.choice(heavy.shape[0], size=max(1, heavy.shape[0] // 40), replace=False)] *= 300.0
    for g_ in (grad, heavy.reshape(grad.shape)):
        args = dict(weight=_t(w), bias=_t(bias), x=_t(x), out=_t(out), grad=_t(g_))
        try:
            eng.debug_variant(8388608)
            pruned = eng.linear_calibrate(**args, **hp)
            again = eng.linear_calibrate(**args, **hp)
            nomemo = eng.linear_calibrate(memoize=False, **args, **hp)
        finally:
            eng.debug_variant(0)
        auto = eng.linear_calibrate(**args, **hp)                  # the engine's own choice (fraction of the weight in the slice)
        full = eng.linear_calibrate(prune=False, **args, **hp)
        torch.cuda.synchronize()
        for k, what in ((0, "w_interval"), (1, "a_interval")):
            assert torch.equal(pruned[k], again[k]), f"pruned search is not run-to-run deterministic ({what})"
            assert torch.equal(pruned[k], full[k]), f"pruned search selected another {what}: {pruned[k].tolist()} vs {full[k].tolist()}"
            assert torch.equal(pruned[k], nomemo[k]), f"pruned search without the pass memo differs ({what})"
            assert torch.equal(auto[k], full[k]), f"adaptive search selected another {what}"


@pytest.mark.parametrize("cfg", [
    dict(b=16, hw=224, oc=768, k=16, channelwise=True, w_bit=8, metric="hessian"),    # ViT-B patch embedding, 768 score blocks
    dict(b=16, hw=224, oc=192, k=16, channelwise=False, w_bit=6, metric="hessian"),   # layer-wise: one block
    dict(b=4, hw=384, oc=128, k=4, channelwise=True, w_bit=8, metric="L2_norm"),      # Swin-B/384 patch embedding, unweighted
], ids=lambda c: f"{'cw' if c['channelwise'] else 'lw'}-{c['metric']}-w{c['w_bit']}-b{c['b']}-{c['hw']}-oc{c['oc']}")
def test_candidate_pruning_is_exact_conv(eng, cfg):
    """The same for the patch-embedding weight search (fp32 planes, a_bit = 32; slices are rows of the im2col GEMM gathered from
    the image, raw_out / raw_grad transposed to rows once): forced, the engine's own choice and unpruned select the same."""
    rng = np.random.default_rng(13)
    b, hw, oc, k = cfg["b"], cfg["hw"], cfg["oc"], cfg["k"]
    w = (rng.standard_normal((oc, 3, k, k)) * 0.05 * np.linspace(0.3, 3.0, oc)[:, None, None, None]).astype(np.float32)
    bias = (rng.standard_normal(oc) * 0.1).astype(np.float32)
    x = rng.standard_normal((b, 3, hw, hw)).astype(np.float32)
    out = torch.nn.functional.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(bias), stride=k).numpy()
    grad = (rng.standard_normal(out.shape) * 1e-3).astype(np.float32)
    heavy = grad.copy()
    mask = rng.random((b, 1) + out.shape[2:]) < 0.03                 # a few pixels carry the weight
    heavy = np.where(mask, heavy * 300.0, heavy).astype(np.float32)
    hp = dict(w_bit=cfg["w_bit"], a_bit=32, metric=cfg["metric"], eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=3,
              stride=(k, k), padding=(0, 0), dilation=(1, 1), channelwise=cfg["channelwise"])
    for g_ in (grad, heavy):
        args = dict(weight=_t(w), bias=_t(bias), x=_t(x), out=_t(out), grad=_t(g_))
        try:
            eng.debug_variant(8388608)
            pruned = eng.conv_calibrate(**args, **hp)
            again = eng.conv_calibrate(**args, **hp)
        finally:
            eng.debug_variant(0)
        auto = eng.conv_calibrate(**args, **hp)
        full = eng.conv_calibrate(prune=False, **args, **hp)
        torch.cuda.synchronize()
        assert torch.equal(pruned[0], again[0]), "pruned conv search is not run-to-run deterministic"
        assert torch.equal(pruned[0], full[0]), f"pruned conv search selected other intervals at {(pruned[0] != full[0]).sum().item()} channels"
        assert torch.equal(auto[0], full[0]), "adaptive conv search selected other intervals"
        assert torch.equal(pruned[1], full[1])


@pytest.mark.parametrize("kind,b,H,S,D,bit,metric", [("qk", 16, 12, 197, 64, 8, "hessian"), ("sv", 16, 12, 197, 64, 8, "hessian"),
                                                      ("qk", 128, 4, 144, 32, 8, "hessian"), ("sv", 64, 4, 144, 32, 6, "L2_norm"),
                                                      ("qk", 9, 3, 250, 64, 8, "L1_norm"), ("qk", 4, 3, 49, 32, 8, "hessian")],
                         ids=["vit-b-qk", "vit-b-sv-sos", "swin-w12-qk", "swin-w12-sv-w6", "250-tokens", "4-images-unpruned"])
def test_candidate_pruning_is_exact_matmul(eng, kind, b, H, S, D, bit, metric):
    """The same for the attention matmuls (per-head score blocks: one hull over the heads; stage A = the first images)."""
    A, B, out, grad = _mk_attention(41, b, H, S, D, kind)
    hp = dict(A_bit=bit, B_bit=bit, metric=metric, eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=3, sos=(kind == "sv"))
    Bt = _t(np.ascontiguousarray(B.transpose(0, 1, 3, 2))).transpose(-2, -1) if kind == "qk" else _t(B)
    heavy = grad.copy()
    heavy[:, :, 0, :] *= 300.0                                    # the class-token query row carries the weight, as in a ViT
    for g_ in (grad, heavy):
        args = dict(A=_t(A), B=Bt, out=_t(out), grad=_t(g_))
        try:
            eng.debug_variant(8388608)
            pruned = eng.matmul_calibrate(**args, **hp)
        finally:
            eng.debug_variant(0)
        auto = eng.matmul_calibrate(**args, **hp)
        full = eng.matmul_calibrate(prune=False, **args, **hp)
        torch.cuda.synchronize()
        for k, what in ((0, "A_interval"), (1, "B_interval"), (2, "split")):
            if pruned[k] is not None:
                assert torch.equal(pruned[k], full[k]), f"pruned search selected another {what}"
                assert torch.equal(auto[k], full[k]), f"adaptive search selected another {what}"


# ------------------------------------------------------------------------------------------------
# cosine on the LDS-DMA sweep (k_sweep2<., EPI_COS>) == cosine on the generic kernel, bit for bit
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [
    dict(kind="linear", b=5, T=61, K=200, N=390, n_V=3, bit=8),
    dict(kind="linear", b=2, T=197, K=768, N=3072, n_V=1, bit=8),       # ViT-B fc1 at 2 images
    dict(kind="linear", b=2, T=197, K=3072, N=768, n_V=1, bit=6),       # ViT-B fc2 geometry (BasePTQ: no twin)
    dict(kind="qk", b=2, H=6, S=197, D=64, bit=8),
], ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_cosine_on_the_fast_sweep_is_bit_identical_to_the_generic_kernel(eng, cfg):
    """The same three sums per sample and 64-feature slab, accumulated in the same order, reduced by the same k_finish_cos: the
    score tables of every pass and the intervals must agree exactly (p4v_debug_set_tuning(12, 7) keeps the generic kernel)."""
    hp = dict(metric="cosine", eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=2)
    if cfg["kind"] == "linear":
        w, bias, x, out, grad = _mk_linear(11, cfg["b"], cfg["T"], cfg["K"], cfg["N"])
        run = lambda: eng.linear_calibrate(weight=_t(w), bias=_t(bias), x=_t(x), out=_t(out), grad=_t(grad), postgelu=False, n_H=1, n_a=1,
                                           n_V=cfg["n_V"], w_bit=cfg["bit"], a_bit=cfg["bit"], want_scores=True, **hp)
    else:
        rng = np.random.default_rng(5)
        A = rng.standard_normal((cfg["b"], cfg["H"], cfg["S"], cfg["D"])).astype(np.float32)
        B = rng.standard_normal((cfg["b"], cfg["H"], cfg["D"], cfg["S"])).astype(np.float32)
        out = A @ B
        grad = (rng.standard_normal(out.shape) * 1e-3).astype(np.float32)
        run = lambda: eng.matmul_calibrate(A=_t(A), B=_t(B), out=_t(out), grad=_t(grad), sos=False, A_bit=cfg["bit"], B_bit=cfg["bit"],
                                           want_scores=True, **hp)
    def kinds(fn):
        eng.stats_reset()
        eng.stats_enable(True)
        try:
            res = fn()
            torch.cuda.synchronize()
            return res, {r["kernel"] for r in eng.stats_launches()}
        finally:
            eng.stats_enable(False)
    try:
        eng.debug_variant(2048)         # (Linear layers with K <= 768 take k_sweep6 by default: the test below)
        fast, k_fast = kinds(run)
        eng.debug_tuning(12, 7)
        generic, k_gen = kinds(run)
    finally:
        eng.debug_tuning(12, 0)
        eng.debug_variant(0)
    assert k_fast == {"k_sweep2"} and k_gen == {"k_sweep<int8>"}, (k_fast, k_gen)
    for a, g_ in zip(fast, generic):
        if a is not None:
            assert torch.equal(a, g_)


# ------------------------------------------------------------------------------------------------
# cosine on the register-stationary sweep (k_sweep6<EPI_COS / EPI_COS_T>, round 6) == cosine on k_sweep2
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [
    dict(b=3, T=50, K=192, N=192, n_V=3, bit=8),         # DeiT-tiny qkv geometry in small: KT = 3, V blocks of one slab
    dict(b=3, T=50, K=192, N=130, n_V=1, bit=8),         # ragged N: the last slab is partly padding
    dict(b=3, T=49, K=256, N=128, n_V=2, bit=6),         # KT = 4
    dict(b=2, T=70, K=384, N=200, n_V=1, bit=8),         # KT = 6
    dict(b=2, T=70, K=512, N=192, n_V=3, bit=8),         # KT = 8
    dict(b=2, T=197, K=768, N=2304, n_V=3, bit=8),       # ViT-B qkv at 2 images: KT = 12, 394 samples = 2 stationary slabs / 7 tiles
    dict(b=2, T=197, K=768, N=3072, n_V=1, bit=8, eq_n=37),   # ViT-B fc1, odd candidate count
    dict(b=1, T=10, K=192, N=64, n_V=1, bit=8, eq_n=1),  # one candidate, one tile
], ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_cosine_on_the_register_stationary_sweep_matches_the_swapped_sweep(eng, cfg):
    """A cosine Linear search with K <= 768 runs as ONE GEMM on k_sweep6 (samples on the MFMA columns: plain in the activation
    search, transposed operands in the weight search) instead of one k_sweep2 GEMM per V block on swapped operands.  Both write
    (dot, |sim|^2, |raw|^2) per (candidate, 64-feature slab, sample) with the same per-lane order of additions and share
    k_finish_cos: the score tables of every pass and the selected intervals must be IDENTICAL (reference linear.py:406-407,
    483-487)."""
    hp = dict(metric="cosine", eq_alpha=0.01, eq_beta=1.2, eq_n=cfg.get("eq_n", 100), search_round=2)
    w, bias, x, out, grad = _mk_linear(13, cfg["b"], cfg["T"], cfg["K"], cfg["N"])
    run = lambda: eng.linear_calibrate(weight=_t(w), bias=_t(bias), x=_t(x), out=_t(out), grad=_t(grad), postgelu=False, n_H=1, n_a=1,
                                       n_V=cfg["n_V"], w_bit=cfg["bit"], a_bit=cfg["bit"], want_scores=True, **hp)
    def kinds(fn):
        eng.stats_reset()
        eng.stats_enable(True)
        try:
            res = fn()
            torch.cuda.synchronize()
            return res, {r["kernel"] for r in eng.stats_launches()}
        finally:
            eng.stats_enable(False)
    six, k_six = kinds(run)
    try:
        eng.debug_variant(2048)
        two, k_two = kinds(run)
    finally:
        eng.debug_variant(0)
    assert k_six == {"k_sweep6"} and k_two == {"k_sweep2"}, (k_six, k_two)
    for a, b_ in zip(six, two):
        if a is not None:
            assert torch.equal(a, b_), float((a.double() - b_.double()).abs().max())


@pytest.mark.parametrize("cfg", [
    dict(b=2, T=197, K=3072, N=768, n_V=1, bit=8),       # ViT-B fc2 geometry (BasePTQ: plain Linear, no twin)
    dict(b=2, T=70, K=1024, N=384, n_V=3, bit=6),        # V blocks of one 128-feature slab each
    dict(b=1, T=50, K=1280, N=160, n_V=1, bit=8, eq_n=7),    # ragged N (a partly padded slab), few candidates
], ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_cosine_on_the_large_k_sweep_matches_the_swapped_sweep(eng, cfg):
    """K >= 1024 (fc2): the cosine search as one GEMM on k_sweep7<0, EPI_COS> (128-feature slabs) against one k_sweep2 GEMM per
    V block on swapped operands (64-feature slabs).  Same integers, another order of the fp32 additions: score tables within
    SCORE_RTOL of each other, a differing selection only at a tie of the tables (reference linear.py:406-407)."""
    from tests.helpers import SCORE_RTOL, TIE_RTOL
    hp = dict(metric="cosine", eq_alpha=0.01, eq_beta=1.2, eq_n=cfg.get("eq_n", 100), search_round=2)
    w, bias, x, out, grad = _mk_linear(17, cfg["b"], cfg["T"], cfg["K"], cfg["N"])
    run = lambda: eng.linear_calibrate(weight=_t(w), bias=_t(bias), x=_t(x), out=_t(out), grad=_t(grad), postgelu=False, n_H=1, n_a=1,
                                       n_V=cfg["n_V"], w_bit=cfg["bit"], a_bit=cfg["bit"], want_scores=True, **hp)
    def kinds(fn):
        eng.stats_reset()
        eng.stats_enable(True)
        try:
            res = fn()
            torch.cuda.synchronize()
            return res, {r["kernel"] for r in eng.stats_launches()}
        finally:
            eng.stats_enable(False)
    seven, k_seven = kinds(run)
    try:
        eng.debug_variant(2048)
        two, k_two = kinds(run)
    finally:
        eng.debug_variant(0)
    assert k_seven == {"k_sweep7"} and k_two == {"k_sweep2"}, (k_seven, k_two)
    w7, a7, s7 = seven[0], seven[1], seven[2]
    w2, a2, s2 = two[0], two[1], two[2]
    # first pass (weight search of round 0) sees the same inputs on both paths: its table must agree to rounding
    t7, t2 = s7.reshape(-1, s7.shape[-2], s7.shape[-1])[0], s2.reshape(-1, s2.shape[-2], s2.shape[-1])[0]
    scale = t2.abs().max().item()
    assert (t7 - t2).abs().max().item() <= SCORE_RTOL * scale, ((t7 - t2).abs().max().item(), scale)
    if not (torch.equal(w7, w2) and torch.equal(a7, a2)):
        # a different selection must be a tie of the FIRST differing table, to TIE_RTOL
        gap = (t2.max(dim=0).values - t2.gather(0, t7.argmax(dim=0, keepdim=True))[0]).abs().max().item()
        assert gap <= TIE_RTOL * scale, (gap, scale)
