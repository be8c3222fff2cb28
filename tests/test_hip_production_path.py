"""GPU: the path bench.py times -- the DEFAULT call of the engine (exact candidate pruning + pass memo, no score tables) --
pinned directly, not by transitivity:

  (a) at BASELINE shapes (ViT-B/224 x 32 images: proj / fc1 / fc2-twin / qkv at 32 x 197 rows, q.k^T and attn.v at
      32 x 12 x 197, the patch embedding at 32 x 224^2) with the gradient profile of a ViT under the reference's KL loss
      (class-token rows carry > 99 % of raw_grad^2, |g| ~ 1e-10): every search pass of the default call against the torch-CPU
      restatement of the reference scored from the engine's own pass input (tests/follow.py), selection = its argmax or a
      near-tie by ITS scores (TIE_RTOL, hard); the pruning counters assert that the three-stage passes ran;
  (b) a whole ViT-B/224 x 32 calibration with pruning on and off: all ~1 340 interval scalars bit-identical;
  (c) four layer cases calibrated BY THE REFERENCE ITSELF at sizes where the pruning engages (tests/golden/prune_*.npz,
      oracle/gen_golden.py::gen_prune_eligible): default call == the reference's intervals (tie-aware by the reference's
      own tables), and the unpruned call's score tables within SCORE_RTOL of the reference's.

Reference: quant_layers/linear.py:455-555, matmul.py:483-576,600-644, conv.py:526-607.
"""
import contextlib
import io

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.follow import _expect_staged, _lookup, follow_conv, follow_linear, follow_matmul
from tests.helpers import (assert_argmax_tie_aware, assert_scores_close, candidate_grid, golden_names, load_golden)

pytestmark = pytest.mark.gpu

PTQ4VIT = dict(metric="hessian", eq_alpha=0.01, eq_beta=1.2, eq_n=100)


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from ptq4vit_amd import engine
    return engine


def vit_like_grad(shape, token_dim, g, scale=1e-10):
    """Noise of the magnitude the reference's KL(pred || raw_pred) gradient has with the class-token rows
    lifted so that they hold > 99 % of grad^2, unevenly over the images (profiles/r3_row_mass.txt: max / mean row mass ~ 2000
    on ViT-B/224 x 32)."""
    grad = torch.randn(shape, generator=g) * scale
    per_image = torch.exp(torch.randn(shape[0], generator=g))
    idx = [slice(None)] * len(shape)
    idx[token_dim] = 0
    view = [shape[0]] + [1] * (len(shape) - 2)
    grad[tuple(idx)] *= 300.0 * per_image.view(view)
    return grad


def _row_mass_stats(grad, rows_dim_end):
    m = (grad.double() ** 2).flatten(rows_dim_end).sum(-1).flatten()
    top = torch.topk(m, max(1, m.numel() // 64)).values.sum() / m.sum()
    return float(m.max() / m.mean()), float(top)


@pytest.mark.parametrize("layer,K,N,n_V,postgelu", [("proj", 768, 768, 1, False), ("fc1", 768, 3072, 1, False),
                                                    ("fc2", 3072, 768, 1, True), ("qkv", 768, 2304, 3, False)])
def test_default_linear_call_at_baseline_shape_follows_the_reference_pass_by_pass(eng, layer, K, N, n_V, postgelu):
    g = torch.Generator().manual_seed({"proj": 11, "fc1": 12, "fc2": 13, "qkv": 14}[layer])
    x = torch.randn(32, 197, K, generator=g)
    if postgelu:
        x = F.gelu(1.5 * x)
    w = torch.nn.init.trunc_normal_(torch.empty(N, K), std=0.02, generator=g) * torch.linspace(0.7, 1.4, N)[:, None]
    b = torch.randn(N, generator=g) * 0.02
    out = F.linear(x, w, b)
    grad = vit_like_grad(out.shape, 1, g)
    peak, top = _row_mass_stats(grad, 2)
    assert peak > 1000 and top > 0.99, (peak, top)
    hp = dict(w_bit=8, a_bit=8, n_V=n_V, postgelu=postgelu, **PTQ4VIT)
    flips, w_iv, a_iv = follow_linear(eng, weight=w, bias=b, x=x, out=out, grad=grad, hp=hp, what=f"ViT-B {layer}")
    print(f"[production] ViT-B {layer} 32 x 197: class-token rows {top:.4f} of the weight (max / mean row mass {peak:.0f}); "
          f"6 passes followed, {flips} near-tie flips; w_interval {w_iv.tolist()[:3]} a_interval {a_iv.tolist()}")


@pytest.mark.parametrize("kind", ["qk", "sv"])
def test_default_matmul_call_at_baseline_shape_follows_the_reference_pass_by_pass(eng, kind):
    g = torch.Generator().manual_seed(21 if kind == "qk" else 22)
    b, H, S, D = 32, 12, 197, 64
    if kind == "qk":
        A = torch.randn(b, H, S, D, generator=g) * torch.linspace(0.5, 2.0, H).view(1, H, 1, 1)
        B = (torch.randn(b, H, S, D, generator=g) * torch.linspace(2.0, 0.5, H).view(1, H, 1, 1)).transpose(-2, -1)
    else:
        A = torch.softmax(torch.randn(b, H, S, S, generator=g) * 3.0, dim=-1)
        B = torch.randn(b, H, S, D, generator=g) * torch.linspace(2.0, 0.5, H).view(1, H, 1, 1)
    out = A @ B
    grad = vit_like_grad(out.shape, 2, g)
    hp = dict(A_bit=8, B_bit=8, **PTQ4VIT)
    flips, A_iv, B_iv, split = follow_matmul(eng, A=A, B=B, out=out, grad=grad, hp=hp, sos=(kind == "sv"), what=f"ViT-B {kind}")
    print(f"[production] ViT-B {'q.k^T' if kind == 'qk' else 'attn.v (split of softmax)'} 32 x 12 x 197: 6 passes followed, "
          f"{flips} near-tie flips; split {None if split is None else float(split)}")


def test_default_conv_call_at_baseline_shape_follows_the_reference(eng):
    g = torch.Generator().manual_seed(23)
    w = torch.nn.init.trunc_normal_(torch.empty(768, 3, 16, 16), std=0.02, generator=g) * torch.linspace(0.5, 2.0, 768).view(-1, 1, 1, 1)
    b = torch.randn(768, generator=g) * 0.02
    x = torch.randn(32, 3, 224, 224, generator=g)
    out = F.conv2d(x, w, b, stride=16)
    # the patch embedding's raw_grad is spread over the pixels (profiles/r3_row_mass.txt: the top 1/8 of the pixel rows hold
    # 0.85 of it): a smooth decay over the pixel rows instead of one heavy row
    grad = torch.randn(out.shape, generator=g) * 1e-10
    decay = torch.exp(-torch.rand(32, 1, 14, 14, generator=g) * 9.0)
    grad = grad * decay
    hp = dict(w_bit=8, a_bit=32, **PTQ4VIT)
    flips, w_iv = follow_conv(eng, weight=w, bias=b, x=x, out=out, grad=grad, stride=16, hp=hp, what="ViT-B patch embedding")
    print(f"[production] ViT-B patch embedding 32 x 224^2: 768 channel searches, {flips} near-tie flips")


def _intervals(wrapped):
    out = {}
    for n, m in wrapped.items():
        out[n] = [torch.as_tensor(getattr(m, a)).detach().clone() for a in ("w_interval", "a_interval", "A_interval", "B_interval", "split")
                  if getattr(m, a, None) is not None and not isinstance(getattr(m, a), (list, tuple))]
    return out


@pytest.mark.parametrize("model,bits,calib,min_staged", [("vit_base_patch16_224", 8, 32, 250), ("vit_small_patch16_224", 8, 32, 150),
                                                         ("vit_base_patch16_224", 6, 32, 150), ("deit_tiny_patch16_224", 8, 16, 50),
                                                         ("swin_tiny_patch4_window7_224", 8, 16, 40)],
                         ids=["vit-b-w8a8-x32", "vit-s-w8a8-x32", "vit-b-w6a6-x32", "deit-tiny-w8a8-x16", "swin-tiny-w8a8-x16-loose-slices"])
def test_whole_network_calibration_is_bit_identical_with_and_without_pruning(eng, model, bits, calib, min_staged):
    """A whole network (the BASELINE headline ViT-B/224 W8A8 PTQ4ViT x 32 images, 74 modules, and three neighbours): the
    calibration bench.py times; the same calibration again (run-to-run determinism of the four search streams); the same with the
    exact candidate pruning switched off (variant 4194304: every candidate over every sample); and under the engine's own
    cross-check (variant 134217728: every pruned pass -- slices, k_bound, second tier -- followed by the full sweep of the SAME
    pass, a differing selection fails the call).  All interval scalars bit-identical."""
    from ptq4vit_amd.configs import PTQ4ViT
    from ptq4vit_amd.utils import models, net_wrap
    from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator
    torch.cuda.empty_cache()
    eng.release_workspace()
    saved = (PTQ4ViT.bit, dict(PTQ4ViT.w_bit), dict(PTQ4ViT.a_bit), dict(PTQ4ViT.A_bit), dict(PTQ4ViT.B_bit))
    PTQ4ViT.bit = bits
    for tab in (PTQ4ViT.w_bit, PTQ4ViT.a_bit, PTQ4ViT.A_bit, PTQ4ViT.B_bit):
        for k in tab:
            tab[k] = bits
    try:
        net = models.get_net(model, seed=0, device="cuda")
        with contextlib.redirect_stdout(io.StringIO()):
            wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
    finally:
        PTQ4ViT.bit = saved[0]
        for tab, old in zip((PTQ4ViT.w_bit, PTQ4ViT.a_bit, PTQ4ViT.A_bit, PTQ4ViT.B_bit), saved[1:]):
            tab.clear()
            tab.update(old)
    images = torch.randn(calib, 3, 224, 224, generator=torch.Generator().manual_seed(0)).cuda()

    class Loader:
        batch_size = calib

        def __iter__(self):
            yield images, None

    def calibrate():
        for m in wrapped.values():
            m.mode = "raw"
        eng.prune_counters(reset=True)
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            HessianQuantCalibrator(net, wrapped, Loader(), sequential=False, batch_size=4).batching_quant_calib()
        torch.cuda.synchronize()
        return _intervals(wrapped), eng.prune_counters(reset=True)

    # Swin has no class token: a slice holds 0.1-0.25 of the metric weight, and only the 128-image configurations are large
    # enough (>= 65 536 sample rows) for the engine to prune on such a slice (round 6).  Tuning key 15 lowers that size limit, so
    # that the LOOSE-slice passes run here -- exactness must not depend on how much of the weight the slice holds.
    if "swin" in model:
        eng.debug_tuning(15, 1024)
    try:
        pruned, c_on = calibrate()
        again, _ = calibrate()
        eng.debug_variant(4194304)
        full, c_off = calibrate()
        eng.debug_variant(134217728)
        checked, c_chk = calibrate()
    finally:
        eng.debug_variant(0)
        eng.debug_tuning(15, 0)
    assert c_chk["staged"] == c_on["staged"], (c_chk, c_on)
    # ViT-B: every executed pass but the `head` Linear's (one row per image: its slice would be the whole layer); the smaller
    # networks keep the passes whose full sweep is cheaper than three staged launches
    assert c_on["staged"] >= min_staged and c_on["not_eligible"] == 0 and (min_staged < 250 or c_on["kept_full_sweep"] <= 6), c_on
    assert c_off["staged"] == 0, c_off
    n = 0
    for name in pruned:
        for a, b, c, d in zip(pruned[name], again[name], full[name], checked[name]):
            assert torch.equal(a, b), f"{name}: pruned calibration is not run-to-run deterministic"
            assert torch.equal(a, c), f"{name}: pruned {a.flatten()[:4].tolist()} vs unpruned {c.flatten()[:4].tolist()}"
            assert torch.equal(a, d), name
            n += a.numel()
    assert n >= 400
    print(f"[production] {model} W{bits}A{bits} x {calib}: {n} interval scalars bit-identical with pruning on ({c_on}), off ({c_off}) and cross-checked")
    del net, wrapped, images
    torch.cuda.empty_cache()
    eng.release_workspace()


# ---- (c) the reference's own run at pruning-eligible sizes ---------------------------------------------------------------
def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _follow_reference_tables(picks, tables, what):
    """picks[i]: candidate indices the engine selected in pass i; tables[i]: the table the REFERENCE fed to argmax in its pass i.
    The passes are comparable as long as both followed the same trajectory: stop at the first pass where they part (which must
    be a near-tie by the reference's own scores).  Returns the number of passes compared and whether the trajectories parted."""
    for i, (idx, tab) in enumerate(zip(picks, tables)):
        tab = np.asarray(tab).reshape(tab.shape[0], -1)
        flips = assert_argmax_tie_aware(idx, tab, what=f"{what} pass {i}")
        if flips:
            return i + 1, True
    return len(picks), False


@pytest.mark.parametrize("name", golden_names("prune_linear_") + golden_names("prune_postgelu_"))
def test_default_linear_call_reproduces_the_reference_at_a_pruning_eligible_size(eng, name):
    from oracle.torch_port import TorchLinear
    g = load_golden(name)
    p = dict(g["params"])
    p.pop("kind"); p.pop("oc")
    R = p.pop("search_round")
    args = dict(weight=_t(g["weight"]).cuda(), bias=_t(g["bias"]).cuda(), x=_t(g["x"]).cuda(), out=_t(g["out"]).cuda(),
                grad=_t(g["grad"]).cuda(), n_H=1, n_a=1, **p)
    # the unpruned call's tables against the reference's (the bar of tests/test_hip_parity.py, at this size)
    w_f, a_f, scores, best = eng.linear_calibrate(search_round=R, want_scores=True, **args)
    torch.cuda.synchronize()
    scores, best = scores.cpu().numpy(), best.cpu().numpy()
    parted = False
    for r in range(R):
        for k, (tab, idx) in enumerate(((scores[r, 0], best[r, 0]), (scores[r, 1][:, :1], best[r, 1][:1]))):
            ref = g["scores"][2 * r + k].reshape(tab.shape[0], -1)
            if not parted:
                assert_scores_close(tab[:, :ref.shape[1]], ref, what=f"{name} round {r} {'wa'[k]}")
                parted = assert_argmax_tie_aware(idx[:ref.shape[1]], ref, what=f"{name} round {r} {'wa'[k]}") > 0
    # the default call, pass by pass against the reference's own tables
    port = TorchLinear(g["weight"], g["bias"], **{k: v for k, v in p.items()})
    _, _, w_c, a_c = port.initial(_t(g["x"]))
    picks = []
    for r in range(1, R + 1):
        eng.prune_counters(reset=True)
        w_iv, a_iv, sc, _ = eng.linear_calibrate(search_round=r, **args)
        torch.cuda.synchronize()
        assert sc is None
        _expect_staged(eng, f"{name} R={r}")
        picks += [_lookup(w_c.numpy(), w_iv.cpu().numpy(), name), _lookup(a_c.numpy()[:, None], a_iv.cpu().numpy(), name)]
    done, parted2 = _follow_reference_tables(picks, g["scores"], name)
    assert torch.equal(w_iv, w_f) and torch.equal(a_iv, a_f), f"{name}: pruned and unpruned calls differ"
    if not parted2:
        np.testing.assert_array_equal(w_iv.cpu().numpy(), g["w_interval"].reshape(-1))
        np.testing.assert_array_equal(a_iv.cpu().numpy(), g["a_interval"].reshape(-1))
    print(f"[production] {name}: {done} of {2 * R} passes on the reference's trajectory"
          f"{' (parted at a near-tie)' if parted2 else ', intervals bit-identical to the reference run'}")


@pytest.mark.parametrize("name", golden_names("prune_matmul_"))
def test_default_matmul_call_reproduces_the_reference_at_a_pruning_eligible_size(eng, name):
    from oracle.torch_port import TorchMatMul
    g = load_golden(name)
    p = dict(g["params"])
    p.pop("kind")
    sos = p.pop("sos")
    R = p.pop("search_round")
    A, B = _t(g["A"]), _t(g["B"])
    Bd = B.cuda() if sos else B.transpose(-2, -1).contiguous().cuda().transpose(-2, -1)      # q.k^T: a transposed view
    args = dict(A=A.cuda(), B=Bd, out=_t(g["out"]).cuda(), grad=_t(g["grad"]).cuda(), sos=sos, **p)
    A_f, B_f, split_f, scores, best = eng.matmul_calibrate(search_round=R, want_scores=True, **args)
    torch.cuda.synchronize()
    scores, best = scores.cpu().numpy(), best.cpu().numpy()
    parted = False
    for r in range(R):
        ta, tb = g["scores"][2 * r], g["scores"][2 * r + 1]
        pairs = [((scores[r, 0][:20, :1], best[r, 0][:1]) if sos else (scores[r, 0], best[r, 0]), ta), ((scores[r, 1], best[r, 1]), tb)]
        for (tab, idx), ref in pairs:
            ref = ref.reshape(tab.shape[0], -1)
            if not parted:
                assert_scores_close(tab, ref, what=f"{name} round {r}")
                parted = assert_argmax_tie_aware(idx, ref, what=f"{name} round {r}") > 0
    port = TorchMatMul(sos=sos, **p)
    _, _, A_c, B_c = port.initial(A, B)
    picks = []
    for r in range(1, R + 1):
        eng.prune_counters(reset=True)
        A_iv, B_iv, split, sc, _ = eng.matmul_calibrate(search_round=r, **args)
        torch.cuda.synchronize()
        assert sc is None
        _expect_staged(eng, f"{name} R={r}")
        if sos:
            i = np.nonzero(np.asarray(port.SPLITS, dtype=np.float32) == np.float32(float(split)))[0]
            assert i.size == 1
            picks.append(i[:1])
        else:
            picks.append(_lookup(A_c.numpy(), A_iv.cpu().numpy(), name))
        picks.append(_lookup(B_c.numpy(), B_iv.cpu().numpy(), name))
    done, parted2 = _follow_reference_tables(picks, g["scores"], name)
    assert torch.equal(A_iv, A_f) and torch.equal(B_iv, B_f), f"{name}: pruned and unpruned calls differ"
    if not parted2:
        np.testing.assert_array_equal(B_iv.cpu().numpy(), g["B_interval"].reshape(-1))
        np.testing.assert_array_equal(A_iv.cpu().numpy(), np.asarray(g["A_interval"]).reshape(-1))
        if sos:
            assert float(split.cpu()) == float(g["split"])
    print(f"[production] {name}: {done} of {2 * R} passes on the reference's trajectory"
          f"{' (parted at a near-tie)' if parted2 else ', intervals bit-identical to the reference run'}")


def test_two_tier_pruning_is_exact_and_engages_on_a_spread_weight_profile(eng):
    """A qkv-like Linear (16 x 197 samples, 768 -> 2304, three score blocks) whose metric weight is spread over the samples --
    per-sample log-normal raw_grad, the 256-row first slice holds between half and 0.9 of it: many candidates survive stage B1's
    bound, the survivors are swept over the second, larger slice (stage A2) before stage B2.  Same intervals as the full sweep of
    every candidate over every sample, and the launch records show the second tier ran."""
    g = torch.Generator().manual_seed(31)
    b, T, K, N = 16, 197, 768, 2304
    x = torch.randn(b, T, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.02 * torch.linspace(0.5, 2.0, N)[:, None]
    bias = torch.randn(N, generator=g) * 0.02
    out = F.linear(x, w, bias)
    grad = torch.randn(out.shape, generator=g) * 1e-10 * torch.exp(1.1 * torch.randn(b, T, 1, generator=g))
    m = (grad.double() ** 2).sum(-1).flatten()
    share = float(torch.topk(m, 256).values.sum() / m.sum())
    assert 0.5 < share < 0.9, share
    args = dict(weight=w.cuda(), bias=bias.cuda(), x=x.cuda(), out=out.cuda(), grad=grad.cuda(), w_bit=8, a_bit=8, n_V=3, n_H=1, n_a=1,
                search_round=3, **PTQ4VIT)
    eng.stats_reset()
    eng.stats_enable(True)
    try:
        two = eng.linear_calibrate(**args)
        torch.cuda.synchronize()
        eng.stats_get()
        stages = [r["stage"] for r in eng.stats_launches()]
    finally:
        eng.stats_enable(False)
    assert "A2" in stages, f"the second tier did not run (first slice holds {share:.2f} of the weight): {sorted(set(stages))}"
    try:
        eng.debug_tuning(13, 1)                      # no second tier
        one = eng.linear_calibrate(**args)
        eng.debug_tuning(13, 0)
        eng.debug_variant(134217728)                 # the engine's own cross-check of every pruned pass against its full sweep
        chk = eng.linear_calibrate(**args)
    finally:
        eng.debug_tuning(13, 0)
        eng.debug_variant(0)
    full = eng.linear_calibrate(prune=False, **args)
    torch.cuda.synchronize()
    for k in (0, 1):
        assert torch.equal(two[k], full[k]) and torch.equal(one[k], full[k]) and torch.equal(chk[k], full[k])
    print(f"[production] two-tier pruning: first slice {share:.2f} of the weight; stages of the run: "
          f"{ {s: stages.count(s) for s in sorted(set(stages))} }")


def test_per_score_block_candidate_ranges_are_exact_and_skip_the_closed_blocks(eng):
    """A ViT-qkv-like Linear (n_V = 3): the q block's metric weight sits in the class-token rows (one survivor: the block is closed
    after the bound), the k / v blocks' weight is spread over every token (flat optima: many survivors).  With per-score-block
    candidate ranges (k_prune_hull's rblk) stages A2 / B2 sweep each block over ITS survivors only.  Same intervals as with the
    ranges switched off (tuning 12 = 9: every block sweeps the hull), as the engine's own cross-check against the full sweep of
    every pass, and as the unpruned call; and the launch records show less executed work in stages A2 / B2."""
    g = torch.Generator().manual_seed(37)
    b, T, K, N = 16, 197, 768, 2304
    x = torch.randn(b, T, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.02 * torch.linspace(0.5, 2.0, N)[:, None]
    bias = torch.randn(N, generator=g) * 0.02
    out = F.linear(x, w, bias)
    grad = torch.randn(out.shape, generator=g) * 1e-10 * torch.exp(0.9 * torch.randn(b, T, 1, generator=g))
    grad[:, 1:, :768] *= 1e-3                        # q block: only the class-token queries reach the classifier
    grad[:, 0, :768] *= 30.0
    args = dict(weight=w.cuda(), bias=bias.cuda(), x=x.cuda(), out=out.cuda(), grad=grad.cuda(), w_bit=8, a_bit=8, n_V=3, n_H=1, n_a=1,
                search_round=3, **PTQ4VIT)

    def staged_ops():
        eng.stats_reset()
        eng.stats_enable(True)
        try:
            res = eng.linear_calibrate(**args)
            torch.cuda.synchronize()
            eng.stats_get()
            recs = eng.stats_launches()
        finally:
            eng.stats_enable(False)
        return res, sum(r["alg_ops"] for r in recs if r["stage"] in ("A2", "B2")), sorted({r["stage"] for r in recs})

    blk, ops_blk, stages = staged_ops()
    assert "B2" in stages, f"no stage B2 ran -- the case does not exercise the ranges: {stages}"
    try:
        eng.debug_tuning(12, 9)                      # per-block ranges off: every block sweeps the hull of all survivors
        hull, ops_hull, _ = staged_ops()
        eng.debug_tuning(12, 0)
        eng.debug_variant(134217728)
        chk = eng.linear_calibrate(**args)
    finally:
        eng.debug_tuning(12, 0)
        eng.debug_variant(0)
    full = eng.linear_calibrate(prune=False, **args)
    torch.cuda.synchronize()
    for k in (0, 1):
        assert torch.equal(blk[k], full[k]) and torch.equal(hull[k], full[k]) and torch.equal(chk[k], full[k])
    assert ops_blk < 0.9 * ops_hull, (ops_blk, ops_hull)
    print(f"[production] per-block ranges: stages A2 + B2 execute {ops_blk / 1e9:.1f} GOP instead of {ops_hull / 1e9:.1f} GOP; stages {stages}")


def test_second_tier_on_a_twin_layer_under_the_cross_check(eng):
    """Advisor finding of round 4: the twin (post-GELU) activation search rebuilds its folded target in place for every pass; both
    slice tiers must re-gather its rows.  A fc2-like twin layer with a spread metric weight and the second tier forced on from two
    survivors (tuning 13 = 2), three rounds, under the engine's cross-check of every pruned pass against its full sweep; and the
    same intervals as the unpruned call."""
    g = torch.Generator().manual_seed(43)
    b, T, K, N = 16, 197, 1024, 768
    x = F.gelu(1.5 * torch.randn(b, T, K, generator=g))
    w = torch.randn(N, K, generator=g) * 0.02
    bias = torch.randn(N, generator=g) * 0.02
    out = F.linear(x, w, bias)
    grad = torch.randn(out.shape, generator=g) * 1e-10 * torch.exp(1.1 * torch.randn(b, T, 1, generator=g))
    args = dict(weight=w.cuda(), bias=bias.cuda(), x=x.cuda(), out=out.cuda(), grad=grad.cuda(), w_bit=8, a_bit=8, n_V=1, n_H=1, n_a=1,
                search_round=3, postgelu=True, **PTQ4VIT)
    try:
        eng.debug_tuning(13, 2)
        eng.stats_reset()
        eng.stats_enable(True)
        try:
            two = eng.linear_calibrate(**args)
            torch.cuda.synchronize()
            eng.stats_get()
            stages = [r["stage"] for r in eng.stats_launches()]
        finally:
            eng.stats_enable(False)
        eng.debug_variant(134217728)
        chk = eng.linear_calibrate(**args)
    finally:
        eng.debug_tuning(13, 0)
        eng.debug_variant(0)
    full = eng.linear_calibrate(prune=False, **args)
    torch.cuda.synchronize()
    for k in (0, 1):
        assert torch.equal(two[k], full[k]) and torch.equal(chk[k], full[k])
    print(f"[production] twin layer, forced second tier: stages { {s_: stages.count(s_) for s_ in sorted(set(stages))} }")


def test_stage_kernels_of_the_pruned_passes_are_the_ones_that_run(eng):
    """The kernels built for the stages of a pruned pass -- k_bound (stage B1 of Linear passes), k_slice_a / k_slice_b (stage A of the
    attention matmuls' A / B searches) -- are on the default path at ViT-B shapes (a silent fall-back to the sweep kernels would keep
    every parity test green and only show in the bench), and the records carry their stage."""
    g = torch.Generator().manual_seed(41)
    b, H, S, D = 8, 12, 197, 64
    runs = {}
    x = torch.randn(b, S, 768, generator=g)
    w = torch.randn(768, 768, generator=g) * 0.02
    out = F.linear(x, w)
    grad = vit_like_grad(out.shape, 1, g)
    runs["linear"] = lambda: eng.linear_calibrate(weight=w.cuda(), bias=None, x=x.cuda(), out=out.cuda(), grad=grad.cuda(), w_bit=8, a_bit=8,
                                                  n_V=1, n_H=1, n_a=1, search_round=1, **PTQ4VIT)
    A = torch.randn(b, H, S, D, generator=g)
    Bk = (torch.randn(b, H, S, D, generator=g)).cuda().transpose(-2, -1)
    o1 = A.cuda() @ Bk
    g1 = vit_like_grad(tuple(o1.shape), 2, g).cuda()
    runs["qk"] = lambda: eng.matmul_calibrate(A=A.cuda(), B=Bk, out=o1, grad=g1, A_bit=8, B_bit=8, search_round=1, sos=False, **PTQ4VIT)
    P = torch.softmax(torch.randn(b, H, S, S, generator=g) * 3.0, dim=-1).cuda()
    V = torch.randn(b, H, S, D, generator=g).cuda()
    o2 = P @ V
    g2 = vit_like_grad(tuple(o2.shape), 2, g).cuda()
    runs["sv"] = lambda: eng.matmul_calibrate(A=P, B=V, out=o2, grad=g2, A_bit=8, B_bit=8, search_round=1, sos=True, **PTQ4VIT)
    seen = {}
    for name, fn in runs.items():
        eng.stats_reset()
        eng.stats_enable(True)
        try:
            fn()
            torch.cuda.synchronize()
            eng.stats_get()
            seen[name] = {(r["kernel"], r["stage"]) for r in eng.stats_launches()}
        finally:
            eng.stats_enable(False)
    assert ("k_bound", "B1") in seen["linear"], seen["linear"]
    assert ("k_slice_a", "A") in seen["qk"] and ("k_slice_b", "A") in seen["qk"], seen["qk"]
    assert ("k_slice_b", "A") in seen["sv"], seen["sv"]
    print(f"[production] stage kernels: {seen}")


# ---- the row selection of the slices (k_topk_rows): exactly the k heaviest rows, each once --------------------------------------
def _topk_reference(mass, k):
    """indices of the k largest entries, ties to the lowest index, ascending; negative / -0.0 masses rank as the lightest"""
    key = mass.detach().cpu().double().clone()
    bits = mass.detach().cpu().view(torch.int32)
    key[bits < 0] = 0.0                                   # sign bit set: lightest (key 0)
    key[torch.isnan(mass.cpu())] = float("inf")           # (positive NaN bit patterns sort above every finite mass)
    out = []
    for row in key:
        order = sorted(range(row.numel()), key=lambda i: (-row[i].item(), i))[:k]
        out.append(sorted(order))
    return torch.tensor(out, dtype=torch.int32)


@pytest.mark.parametrize("segs,n,k,kind", [(1, 6304, 512, "lognormal"), (1, 6304, 128, "ties"), (1, 1000, 999, "lognormal"), (1, 300, 1, "lognormal"),
                                           (384, 197, 16, "cls"), (7, 1025, 256, "ties"), (1, 70000, 1280, "lognormal"), (1, 4096, 256, "zeros"),
                                           (3, 5000, 320, "negatives"), (1, 1200000, 4864, "lognormal"), (1, 2048, 2048, "lognormal")])
def test_row_selection_takes_exactly_the_heaviest_rows_once(eng, segs, n, k, kind):
    g = torch.Generator().manual_seed(segs * 1000 + n + k)
    if kind == "lognormal":
        m = torch.exp(4.0 * torch.randn(segs, n, generator=g)) * 1e-18
    elif kind == "ties":
        m = torch.randint(0, 6, (segs, n), generator=g).float() * 0.37          # many equal masses around the threshold
    elif kind == "cls":
        m = torch.rand(segs, n, generator=g) * 1e-22
        m[:, 0] = 3e-17                                                         # one class-token row per segment
    elif kind == "zeros":
        m = torch.zeros(segs, n)
        m[0, torch.randperm(n, generator=g)[:100]] = torch.rand(100, generator=g)    # fewer non-zero rows than k
    else:
        m = torch.randn(segs, n, generator=g)                                    # signed: negative masses are the lightest
        m[:, ::7] = -0.0
    idx = eng.debug_topk_rows(m.cuda().contiguous(), k).cpu()
    assert idx.shape == (segs, k)
    assert bool((idx[:, 1:] > idx[:, :-1]).all()) or k == 1, "indices must be strictly ascending (no row twice)"
    assert int(idx.min()) >= 0 and int(idx.max()) < n
    if n <= 70000:
        ref = _topk_reference(m, k)
        assert torch.equal(idx, ref)
    else:                                                 # large segment: the set is checked through its threshold
        key = m.clone()
        key[m.view(torch.int32) < 0] = 0.0
        sel = torch.zeros(segs, n, dtype=torch.bool)
        sel.scatter_(1, idx.long(), True)
        for s_ in range(segs):
            inside, outside = key[s_][sel[s_]], key[s_][~sel[s_]]
            assert inside.numel() == k and float(inside.min()) >= float(outside.max())
