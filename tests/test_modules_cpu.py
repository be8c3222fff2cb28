"""Host-side mirror of the reference's module / wrapper / config / calibrator API (no GPU needed)."""
import json

import numpy as np
import pytest
import torch

from tests.helpers import golden_names, load_golden

from ptq4vit_amd.configs import BasePTQ, PTQ4ViT
from ptq4vit_amd.quant_layers.conv import BatchingEasyQuantConv2d, ChannelwiseBatchingQuantConv2d, MinMaxQuantConv2d
from ptq4vit_amd.quant_layers.linear import (MinMaxQuantLinear, PostGeluPTQSLBatchingQuantLinear,
                                             PTQSLBatchingQuantLinear, PTQSLQuantLinear)
from ptq4vit_amd.quant_layers.matmul import (MinMaxQuantMatMul, PTQSLBatchingQuantMatMul,
                                             SoSPTQSLBatchingQuantMatMul)
from ptq4vit_amd.utils import models, net_wrap, quant_calib


def test_config_factory_types_and_kwargs():
    qkv = PTQ4ViT.get_module("qlinear_qkv", 48, 144)
    assert type(qkv) is PTQSLBatchingQuantLinear and qkv.n_V == 3 and qkv.metric == "hessian"
    assert (qkv.eq_alpha, qkv.eq_beta, qkv.eq_n, qkv.search_round) == (0.01, 1.2, 100, 3)
    assert type(PTQ4ViT.get_module("qlinear_MLP_2", 192, 48)) is PostGeluPTQSLBatchingQuantLinear
    assert type(PTQ4ViT.get_module("qmatmul_scorev")) is SoSPTQSLBatchingQuantMatMul
    assert type(PTQ4ViT.get_module("qmatmul_qk")) is PTQSLBatchingQuantMatMul
    conv = PTQ4ViT.get_module("qconv", 3, 48, (8, 8), (8, 8), (0, 0), (1, 1), 1, True, "zeros")
    assert type(conv) is ChannelwiseBatchingQuantConv2d and conv.a_bit == 32 and conv.n_V == 48
    assert type(BasePTQ.get_module("qconv", 3, 48, (8, 8), (8, 8), (0, 0), (1, 1), 1, True, "zeros")) is BatchingEasyQuantConv2d
    assert type(BasePTQ.get_module("qlinear_MLP_2", 192, 48)) is PTQSLBatchingQuantLinear
    assert BasePTQ.get_module("qlinear_proj", 48, 48).metric == "cosine"


def test_wrap_order_and_shared_parameters():
    net = models.get_net("vit_tiny_patch16_224", depth=2, device="cpu", img_size=32, patch_size=8, embed_dim=48, num_heads=3, num_classes=10)
    w0 = net.blocks[0].attn.qkv.weight
    wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
    names = list(wrapped)
    assert names[:7] == ["patch_embed.proj", "blocks.0.attn.qkv", "blocks.0.attn.proj", "blocks.0.attn.matmul1",
                         "blocks.0.attn.matmul2", "blocks.0.mlp.fc1", "blocks.0.mlp.fc2"] and names[-1] == "head"
    assert len(wrapped) == 14 and all(m.mode == "raw" for m in wrapped.values())
    assert net.blocks[0].attn.qkv.weight.data_ptr() == w0.data_ptr()      # storage shared with the float module
    x = torch.randn(2, 3, 32, 32)
    with torch.no_grad():
        y = net(x)
    assert y.shape == (2, 10)


def test_mode_dispatch_and_errors():
    m = MinMaxQuantLinear(4, 4)
    m.mode = "bogus"
    with pytest.raises(NotImplementedError):
        m(torch.zeros(1, 4))
    with pytest.raises(AssertionError):
        MinMaxQuantLinear(4, 4, bias_bit=8)
    mm = MinMaxQuantMatMul()
    mm.mode = "bogus"
    with pytest.raises(NotImplementedError):
        mm(torch.zeros(1, 1, 2, 2), torch.zeros(1, 1, 2, 2))
    assert not hasattr(PTQSLBatchingQuantLinear(4, 4), "calibrated")


def test_calibration_without_gpu_fails_loudly():
    """The calibration path has no CPU fallback: without a visible GPU it must raise, not silently compute."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = PTQSLBatchingQuantLinear(8, 8, metric="L2_norm")
    m.raw_input, m.raw_out, m.raw_grad = torch.randn(2, 3, 8), torch.randn(2, 3, 8), None
    with pytest.raises(RuntimeError, match="no CPU fallback|needs an MI355X"):
        m.calibration_step2()


def test_per_pass_methods_without_gpu_fail_loudly():
    """The reference's per-pass methods (_initialize_intervals, _search_best_*_interval) are GPU passes too: no GPU,
    no result -- never a silent CPU computation."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lin = PTQSLBatchingQuantLinear(8, 8, metric="L2_norm")
    lin.raw_input, lin.raw_out, lin.raw_grad = torch.randn(2, 3, 8), torch.randn(2, 3, 8), None
    mm = PTQSLBatchingQuantMatMul(metric="L2_norm")
    mm.raw_input, mm.raw_out, mm.raw_grad = [torch.randn(2, 2, 3, 4), torch.randn(2, 2, 4, 3)], torch.randn(2, 2, 3, 3), None
    conv = ChannelwiseBatchingQuantConv2d(3, 4, 2, stride=2, metric="L2_norm")
    conv.raw_input, conv.raw_out, conv.raw_grad = torch.randn(2, 3, 4, 4), torch.randn(2, 4, 2, 2), None
    for m in (lin, mm, conv):
        with pytest.raises(RuntimeError, match="no CPU fallback|needs an MI355X"):
            m._initialize_intervals()
    cands = torch.ones(101, 1, 1, 1, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback|needs an MI355X"):
        lin.w_interval, lin.a_interval = torch.ones(1, 1, 1, 1), torch.ones(1, 1)
        lin._search_best_w_interval(cands)


@pytest.mark.parametrize("name", ["linear_qkv_hessian_w8a8", "postgelu_hessian_w6a6", "linear_blocks_nH2_na2"])
def test_linear_quant_forward_matches_reference(name):
    """quant_forward (reference linear.py:62-67,601-607) with the reference's calibrated intervals."""
    g = load_golden(name)
    p = dict(g["params"])
    p.pop("kind")
    oc = p.pop("oc")
    cls = PostGeluPTQSLBatchingQuantLinear if p.pop("postgelu") else PTQSLBatchingQuantLinear
    m = cls(g["x"].shape[-1], oc, bias="bias" in g, **p)
    m.weight.data = torch.from_numpy(g["weight"])
    if "bias" in g:
        m.bias.data = torch.from_numpy(g["bias"])
    m.w_interval = torch.from_numpy(g["w_interval"])
    m.a_interval = torch.from_numpy(g["a_interval"])
    m.calibrated, m.mode = True, "quant_forward"
    with torch.no_grad():
        y = m(torch.from_numpy(g["x"]))
    np.testing.assert_allclose(y.numpy(), g["quant_forward"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["matmul_qk_hessian_w8a8", "matmul_sos_hessian_w8a8"])
def test_matmul_quant_forward_matches_reference(name):
    g = load_golden(name)
    p = dict(g["params"])
    p.pop("kind")
    sos = p.pop("sos")
    m = (SoSPTQSLBatchingQuantMatMul if sos else PTQSLBatchingQuantMatMul)(**p)
    A, B = torch.from_numpy(g["A"]), torch.from_numpy(g["B"])
    m.n_G_A = m.n_G_B = A.shape[1]
    m._get_padding_parameters(A, B)
    m.B_interval = torch.from_numpy(g["B_interval"])
    m.A_interval = torch.from_numpy(np.asarray(g["A_interval"]))
    if sos:
        m.split = torch.tensor(float(g["split"]))
    m.calibrated, m.mode = True, "quant_forward"
    with torch.no_grad():
        y = m(A, B)
    np.testing.assert_allclose(y.numpy(), g["quant_forward"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["conv_channelwise_hessian", "conv_layerwise_cosine"])
def test_conv_quant_forward_matches_reference(name):
    g = load_golden(name)
    p = dict(g["params"])
    p.pop("kind")
    st = p.pop("stride")
    cls = ChannelwiseBatchingQuantConv2d if p.pop("channelwise") else BatchingEasyQuantConv2d
    oc, ic, k, _ = g["weight"].shape
    m = cls(ic, oc, k, st, **p)
    m.weight.data, m.bias.data = torch.from_numpy(g["weight"]), torch.from_numpy(g["bias"])
    m.w_interval = torch.from_numpy(np.asarray(g["w_interval"]))
    m.a_interval = torch.from_numpy(np.asarray(g["a_interval"]))
    m.calibrated, m.mode = True, "quant_forward"
    with torch.no_grad():
        y = m(torch.from_numpy(g["x"]))
    np.testing.assert_allclose(y.numpy(), g["quant_forward"], rtol=1e-5, atol=1e-6)


def _mini_net(kw):
    return models.get_net("vit_tiny_patch16_224", seed=0, device="cpu", **kw)


class _Loader:
    def __init__(self, images):
        self.images, self.batch_size = images, images.shape[0]

    def __iter__(self):
        yield self.images, torch.zeros(self.images.shape[0], dtype=torch.long)


@pytest.mark.parametrize("budget", [1 << 40, 1])
def test_one_pass_capture_equals_reference_capture(budget):
    """Capture contract (reference quant_calib.py:309-356): raw_input / raw_out / raw_grad cached by ONE set of
    passes hooking all modules (or groups of modules under a cache budget) are bit-identical to what the
    reference's per-module passes cached (fixture made by oracle/gen_golden.py from the reference)."""
    g = np.load("tests/golden/minivit_ptq4vit.npz", allow_pickle=False)
    kw = json.loads(str(g["model_kwargs"]))
    net = _mini_net(kw)
    wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
    assert list(wrapped) == [str(n) for n in g["names"]]
    seen = {}
    for n, m in wrapped.items():
        def rec(_m=m, _n=n):
            ri = _m.raw_input
            seen[_n] = ([t.clone() for t in ri] if isinstance(ri, list) else ri.clone(), _m.raw_out.clone(), _m.raw_grad.clone())
            _m.calibrated = True
        m.calibration_step2 = rec
    cal = quant_calib.HessianQuantCalibrator(net, wrapped, _Loader(torch.from_numpy(g["images"])), sequential=False,
                                             batch_size=4, cache_budget_bytes=budget,
                                             capture_batch_size=4)      # the reference's passes: bit-identical
    cal.batching_quant_calib()
    assert all(m.mode == "quant_forward" for m in wrapped.values())
    for n in wrapped:
        key = n.replace(".", "__")
        ri, ro, rg = seen[n]
        if isinstance(ri, list):
            np.testing.assert_array_equal(ri[0].numpy(), g[f"{key}::A"])
            np.testing.assert_array_equal(ri[1].numpy(), g[f"{key}::B"])
        else:
            np.testing.assert_array_equal(ri.numpy(), g[f"{key}::x"])
        np.testing.assert_array_equal(ro.numpy(), g[f"{key}::out"])
        np.testing.assert_array_equal(rg.numpy(), g[f"{key}::grad"])


@pytest.mark.parametrize("n_images", [8, 7])
def test_large_capture_passes_equal_the_reference_sub_batches(n_images):
    """`capture_batch_size` (opt-in: fewer, larger capture passes; per-sample KL weights 1 / rows of the reference
    sub-batch) records the tensors of the reference's passes of `batch_size` images (quant_calib.py:309-356, loss :333-339)
    up to the rounding of differently blocked GEMMs -- also when the last reference sub-batch is ragged (7 images at
    batch_size 4: its three samples get gradients divided by 3, not 4).  The target distribution here is NOT the network's
    own prediction (for which the gradients are rounding noise, see HessianQuantCalibrator.__init__) but a fixed random
    one, so that raw_grad is a well-defined quantity."""
    g = np.load("tests/golden/minivit_ptq4vit.npz", allow_pickle=False)
    kw = json.loads(str(g["model_kwargs"]))
    net = _mini_net(kw)
    wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
    images = torch.from_numpy(g["images"])
    images = torch.cat([images, images.flip(0)])[:n_images]
    target = torch.softmax(torch.randn(n_images, kw["num_classes"], generator=torch.Generator().manual_seed(3)), dim=-1)
    caps = []
    for cbs in (None, 8):
        cal = quant_calib.HessianQuantCalibrator(net, wrapped, _Loader(images), sequential=False, batch_size=4,
                                                 capture_batch_size=cbs)
        assert cal._capture_bs() == (8 if cbs else 4)
        cal._capture(list(wrapped), target, True)
        cap = {}
        for n, m in wrapped.items():
            ri = m.raw_input if isinstance(m.raw_input, list) else [m.raw_input]
            cap[n] = [t.clone() for t in ri] + [m.raw_out.clone(), m.raw_grad.clone()]
        caps.append(cap)
    for n in caps[0]:
        for a, b in zip(caps[0][n], caps[1][n]):
            assert a.shape == b.shape
            scale = float(a.abs().max())
            assert scale > 0 and float((a - b).abs().max()) <= 2e-6 * scale, n


def test_fold_bn_into_conv_matches_conv_then_bn():
    """utils/net_wrap.fold_bn_into_conv (reference net_wrap.py:8-36): conv' == bn(conv), affine or not, bias or not."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 3, 9, 9, generator=g)
    for affine in (True, False):
        for bias in (True, False):
            conv = torch.nn.Conv2d(3, 5, 3, bias=bias)
            bn = torch.nn.BatchNorm2d(5, affine=affine).eval()
            bn.running_mean.copy_(torch.randn(5, generator=g))
            bn.running_var.copy_(torch.rand(5, generator=g) + 0.5)
            if affine:
                bn.weight.data.copy_(torch.randn(5, generator=g))
                bn.bias.data.copy_(torch.randn(5, generator=g))
            with torch.no_grad():
                want = bn(conv(x))
                net_wrap.fold_bn_into_conv(conv, bn)
                got = conv(x)
            assert conv.bias is not None
            torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("metric", ["L1_norm", "L2_norm", "linear_weighted_L2_norm", "square_weighted_L2_norm",
                                    "hessian", "cosine"])
def test_get_similarity_matches_the_oracle(metric):
    """the public `_get_similarity` helper of the module classes (reference linear.py:399-424,
    matmul.py:442-481, conv.py:322-351 / 498-524) against the oracle's restatement."""
    from oracle import ptq4vit_oracle as orc
    g = torch.Generator().manual_seed(3)
    raw = torch.randn(2, 5, 1, 3, 8, generator=g)
    sim = raw + 0.05 * torch.randn(2, 5, 4, 3, 8, generator=g)
    grad = 1e-3 * torch.randn(2, 5, 1, 3, 8, generator=g)
    want = orc.similarity_lastdim(raw.numpy(), sim.numpy(), metric, grad.numpy())
    lin = PTQSLBatchingQuantLinear(8, 24, metric=metric)
    np.testing.assert_allclose(lin._get_similarity(raw, sim, metric, raw_grad=grad).numpy(), want, rtol=2e-6, atol=1e-9)
    mm = PTQSLBatchingQuantMatMul(metric=metric)
    np.testing.assert_allclose(mm._get_similarity(raw, sim, metric, dim=-1, raw_grad=grad).numpy(), want, rtol=2e-6, atol=1e-9)
    easy = BatchingEasyQuantConv2d(3, 8, 4, metric=metric)
    np.testing.assert_allclose(easy._get_similarity(raw, sim, metric, dim=-1, raw_grad=grad).numpy(), want, rtol=2e-6, atol=1e-9)
    # channel-wise conv: tensors (b, p, oc, fh, fw); cosine over the pixels of one (image, channel), else element-wise
    raw_c, sim_c, grad_c = raw[:, 0].unsqueeze(1)[:, :, 0], sim[:, 0], grad[:, 0].unsqueeze(1)[:, :, 0]   # (2,1,3,8)->5-D below
    raw_c = raw_c.reshape(2, 1, 3, 2, 4); sim_c = sim_c.reshape(2, 4, 3, 2, 4); grad_c = grad_c.reshape(2, 1, 3, 2, 4)
    cw = ChannelwiseBatchingQuantConv2d(3, 3, 4, metric=metric)
    got = cw._get_similarity(raw_c, sim_c, metric, raw_grad=grad_c).numpy()
    if metric == "cosine":
        want_c = orc._cosine(raw_c.reshape(2, 1, 3, 8).numpy(), sim_c.reshape(2, 4, 3, 8).numpy(), -1).reshape(2, 4, 3, 1, 1)
    else:
        want_c = orc.elementwise_similarity(raw_c.numpy(), sim_c.numpy(), metric, grad_c.numpy())
    np.testing.assert_allclose(got, want_c, rtol=2e-6, atol=1e-9)


def test_capture_cache_layout_helpers():
    """quant_calib._cache_like / _same_dense_block (the capture's one-launch append): a cache keeps the hooked tensor's memory
    layout exactly when a sub-batch piece is then ONE dense block of it; everything else falls back to a contiguous cache and
    torch's copy."""
    from ptq4vit_amd.utils.quant_calib import _cache_like, _same_dense_block
    k = torch.randn(4, 3, 7, 5)
    cases = {
        "contiguous": (torch.randn(4, 7, 5), True),
        "k.transpose(-2, -1) (batch-major dense permutation)": (k.transpose(-2, -1), True),
        "gapped view": (torch.randn(4, 7, 10)[:, :, :5], False),
        "batch not outermost in memory": (torch.randn(7, 4, 5).transpose(0, 1), False),
        "1-D": (torch.randn(6), True),
    }
    for what, (t, dense) in cases.items():
        c = _cache_like(t, 3)
        assert tuple(c.shape) == (3 * t.shape[0],) + tuple(t.shape[1:]), what
        assert _same_dense_block(c, t) == dense, what
        if dense:      # a flat copy of t's storage block IS the copy of piece 1
            n = t.numel()
            c.as_strided((n,), (1,), storage_offset=n).copy_(t.as_strided((n,), (1,), storage_offset=t.storage_offset()))
            assert torch.equal(c[t.shape[0]:2 * t.shape[0]], t), what
    assert not _same_dense_block(_cache_like(k.half(), 2), k.half())          # the append kernel moves 4-byte elements


def test_committed_bench_line_keeps_the_contract():
    """profiles/r*_bench.json is a line bench.py printed on the MI355X: the fields the driver and the judge read."""
    import glob
    path = sorted(glob.glob("profiles/r*_bench.json"))[-1]
    d = json.loads(open(path).read())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "layers/s" and d["higher_is_better"] is True and d["vs_baseline"] is None and "workload" in d["config"]
    assert abs(d["value"] - 74 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None or r["traffic"] > 0
    assert abs(r["achieved"] - r["ops_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == "layers/s" and c["sample"]
