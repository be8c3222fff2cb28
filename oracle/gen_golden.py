"""Generate golden fixtures by IMPORTING the reference (build container only).

Runs the reference's own ``calibration_step2`` (read-only import from
/root/reference, CPU, ``.cuda()`` shimmed to identity -- SURVEY.md App. C) on
small seeded layers and stores inputs + outputs as ``tests/golden/*.npz``.
Nothing from the reference is copied: the fixtures are data (inputs, expected
intervals, expected score tables).  The reference does not exist on the GPU
box, so this script is never run there; the committed ``.npz`` files travel.

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _install_shims():
    sys.path.insert(0, REF)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self


class ArgmaxRecorder:
    """Capture every score table the reference feeds to argmax (the search decisions)."""

    def __init__(self):
        self.tables = []
        self._t_argmax = torch.Tensor.argmax
        self._f_argmax = torch.argmax

    def __enter__(self):
        rec = self

        def t_argmax(self_, *a, **k):
            rec.tables.append(self_.detach().clone().numpy())
            return rec._t_argmax(self_, *a, **k)

        def f_argmax(inp, *a, **k):
            rec.tables.append(inp.detach().clone().numpy())
            return rec._f_argmax(inp, *a, **k)

        torch.Tensor.argmax = t_argmax
        torch.argmax = f_argmax
        return self

    def __exit__(self, *exc):
        torch.Tensor.argmax = self._t_argmax
        torch.argmax = self._f_argmax


def _save(name, params, arrays, tables):
    os.makedirs(OUT, exist_ok=True)
    payload = {k: np.asarray(v) for k, v in arrays.items()}
    for i, t in enumerate(tables):
        payload[f"scores_{i:02d}"] = t
    payload["params"] = np.array(json.dumps(params))
    payload["n_scores"] = np.array(len(tables))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **payload)
    print(f"wrote {name}.npz  ({len(tables)} score tables)")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


# --------------------------------------------------------------------------- #
def _cls_heavy(grad, token_dim, g):
    """raw_grad of a ViT under the reference's KL loss: the classifier reads the class token only, so the rows of token 0
    carry almost all of grad^2 (measured on ViT-B: > 99 %), unevenly over the images.  Scales the class-token rows of a
    noise gradient by 300 x a per-image factor (in place on a clone)."""
    grad = grad.clone()
    per_image = torch.exp(torch.randn(grad.shape[0], generator=g))
    idx = [slice(None)] * grad.dim()
    idx[token_dim] = 0
    shape = [grad.shape[0]] + [1] * (grad[tuple(idx)].dim() - 1)
    grad[tuple(idx)] = grad[tuple(idx)] * 300.0 * per_image.view(shape)
    return grad


def gen_linear(name, *, shape_x, oc, postgelu=False, grad_scale=1e-3, seed=0, bias=True, cls_heavy=False, store_qf=True, **kw):
    from quant_layers.linear import PTQSLBatchingQuantLinear, PostGeluPTQSLBatchingQuantLinear

    g = torch.Generator().manual_seed(seed)
    ic = shape_x[-1]
    w = torch.randn(oc, ic, generator=g) * 0.05
    # per-row-block magnitude spread so the n_V blocks get different intervals
    w = w * torch.linspace(0.5, 2.0, oc).view(-1, 1)
    b = torch.randn(oc, generator=g) * 0.1 if bias else None
    x = torch.randn(*shape_x, generator=g)
    if postgelu:
        x = F.gelu(1.5 * x)
    out = F.linear(x, w, b)
    grad = torch.randn(out.shape, generator=g) * grad_scale
    if cls_heavy:
        grad = _cls_heavy(grad, 1, g)
    cls = PostGeluPTQSLBatchingQuantLinear if postgelu else PTQSLBatchingQuantLinear
    m = cls(ic, oc, bias=bias, **kw)
    m.weight.data = w.clone()
    if bias:
        m.bias.data = b.clone()
    m.raw_input, m.raw_out, m.raw_grad = x.clone(), out.clone(), grad.clone()
    with torch.no_grad(), ArgmaxRecorder() as rec:
        m.calibration_step2()
    m.mode = "quant_forward"
    with torch.no_grad():
        qf = m(x)
    arrays = dict(weight=w.numpy(), x=x.numpy(), out=out.numpy(), grad=grad.numpy(),
                  w_interval=m.w_interval.numpy(), a_interval=m.a_interval.numpy(),
                  quant_forward=qf.numpy(),
                  calib=np.array([m.calib_size, m.calib_batch_size, m.parallel_eq_n]))
    if not store_qf:                                  # the larger fixtures: the output is as big as raw_out
        arrays["quant_forward"] = qf.numpy()[:1, :8]
    if bias:
        arrays["bias"] = b.numpy()
    _save(name, dict(kind="linear", postgelu=postgelu, oc=oc, **kw), arrays, rec.tables)


def gen_matmul(name, *, b, H, d1, d2, d3, sos=False, grad_scale=1e-3, seed=0, cls_heavy=False, store_qf=True, **kw):
    from quant_layers.matmul import PTQSLBatchingQuantMatMul, SoSPTQSLBatchingQuantMatMul

    g = torch.Generator().manual_seed(seed)
    if sos:
        A = torch.softmax(torch.randn(b, H, d1, d2, generator=g) * 3.0, dim=-1)
    else:
        A = torch.randn(b, H, d1, d2, generator=g) * torch.linspace(0.5, 2.0, H).view(1, H, 1, 1)
    # B is handed over as a transposed view in the q.k^T case (utils/models.py:16)
    Bm = (torch.randn(b, H, d3, d2, generator=g) * torch.linspace(2.0, 0.5, H).view(1, H, 1, 1)).transpose(-2, -1)
    out = A @ Bm
    grad = torch.randn(out.shape, generator=g) * grad_scale
    if cls_heavy:
        grad = _cls_heavy(grad, 2, g)                 # the query row of the class token
    cls = SoSPTQSLBatchingQuantMatMul if sos else PTQSLBatchingQuantMatMul
    m = cls(**kw)
    m.raw_input, m.raw_out, m.raw_grad = [A.clone(), Bm.clone()], out.clone(), grad.clone()
    with torch.no_grad(), ArgmaxRecorder() as rec:
        m.calibration_step2()
    m.mode = "quant_forward"
    with torch.no_grad():
        qf = m(A, Bm)
    arrays = dict(A=A.numpy(), B=Bm.contiguous().numpy(), out=out.numpy(), grad=grad.numpy(),
                  A_interval=np.asarray(m.A_interval), B_interval=m.B_interval.numpy(),
                  quant_forward=qf.numpy())
    if not store_qf:
        arrays["quant_forward"] = qf.numpy()[:1, :1, :8]
    if sos:
        arrays["split"] = np.asarray(m.split)
    _save(name, dict(kind="matmul", sos=sos, **kw), arrays, rec.tables)


def gen_conv(name, *, b, ic, hw, oc, k, stride, channelwise=True, grad_scale=1e-3, seed=0, **kw):
    from quant_layers.conv import BatchingEasyQuantConv2d, ChannelwiseBatchingQuantConv2d

    g = torch.Generator().manual_seed(seed)
    w = torch.randn(oc, ic, k, k, generator=g) * 0.05 * torch.linspace(0.5, 2.0, oc).view(-1, 1, 1, 1)
    bias = torch.randn(oc, generator=g) * 0.1
    x = torch.randn(b, ic, hw, hw, generator=g)
    out = F.conv2d(x, w, bias, stride)
    grad = torch.randn(out.shape, generator=g) * grad_scale
    cls = ChannelwiseBatchingQuantConv2d if channelwise else BatchingEasyQuantConv2d
    m = cls(ic, oc, k, stride, **kw)
    m.weight.data = w.clone()
    m.bias.data = bias.clone()
    m.raw_input, m.raw_out, m.raw_grad = x.clone(), out.clone(), grad.clone()
    with torch.no_grad(), ArgmaxRecorder() as rec:
        m.calibration_step2()
    m.mode = "quant_forward"
    with torch.no_grad():
        qf = m(x)
    arrays = dict(weight=w.numpy(), bias=bias.numpy(), x=x.numpy(), out=out.numpy(), grad=grad.numpy(),
                  w_interval=np.asarray(m.w_interval), a_interval=np.asarray(m.a_interval),
                  quant_forward=qf.numpy())
    _save(name, dict(kind="conv", channelwise=channelwise, stride=stride, **kw), arrays, rec.tables)


def gen_ptqsl_conv(name, *, b, ic, hw, oc, k, stride, grad_scale=1e-3, seed=0, **kw):
    """PTQSLQuantConv2d's own (non-batching) search, conv.py:126-277: calibration_step2(x)."""
    from quant_layers.conv import PTQSLQuantConv2d

    g = torch.Generator().manual_seed(seed)
    w = torch.randn(oc, ic, k, k, generator=g) * 0.05 * torch.linspace(0.5, 2.0, oc).view(-1, 1, 1, 1)
    bias = torch.randn(oc, generator=g) * 0.1
    x = torch.randn(b, ic, hw, hw, generator=g)
    out = F.conv2d(x, w, bias, stride)
    grad = torch.randn(out.shape, generator=g) * grad_scale
    m = PTQSLQuantConv2d(ic, oc, k, stride, parallel_eq_n=10, **kw)
    m.weight.data = w.clone()
    m.bias.data = bias.clone()
    m.raw_input, m.raw_out = x.clone(), out.clone()
    m.raw_grad = grad.clone() if kw.get("metric") == "hessian" else None
    with torch.no_grad(), ArgmaxRecorder() as rec:
        qf = m.calibration_step2(x)
    arrays = dict(weight=w.numpy(), bias=bias.numpy(), x=x.numpy(), out=out.numpy(), grad=grad.numpy(),
                  w_interval=np.asarray(m.w_interval), a_interval=np.asarray(m.a_interval), quant_forward=qf.numpy())
    _save(name, dict(kind="ptqsl_conv", stride=stride, **kw), arrays, rec.tables)


PTQ4VIT = dict(metric="hessian", eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=3)
BASEPTQ = dict(metric="cosine", eq_alpha=0.5, eq_beta=1.2, eq_n=100, search_round=1)


def gen_ptqsl_convs():
    # ---- PTQSLQuantConv2d's own search (conv.py:126-277) --------
    gen_ptqsl_conv("ptqslconv_hessian_v2h2", b=4, ic=3, hw=32, oc=12, k=8, stride=8, w_bit=8, a_bit=8, n_V=2, n_H=2, seed=40,
                   metric="hessian", eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=2)
    gen_ptqsl_conv("ptqslconv_l2_v3h1_overlap", b=3, ic=3, hw=20, oc=12, k=5, stride=3, w_bit=6, a_bit=6, n_V=3, n_H=1, seed=41,
                   metric="L2_norm", eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=1)
    gen_ptqsl_conv("ptqslconv_cosine_v1h3", b=4, ic=3, hw=32, oc=12, k=8, stride=8, w_bit=8, a_bit=8, n_V=1, n_H=3, seed=42,
                   metric="cosine", eq_alpha=0.5, eq_beta=1.2, eq_n=100, search_round=1)


def main(only=None):
    _install_shims()
    os.chdir(REF)
    if only == "minivit":
        return gen_mini_vit()
    if only == "ptqslconv":
        return gen_ptqsl_convs()
    # ---- Linear (quant_layers/linear.py:349-642) ---------------------------------
    gen_linear("linear_qkv_hessian_w8a8", shape_x=(4, 13, 48), oc=36, n_V=3, w_bit=8, a_bit=8, **PTQ4VIT)
    gen_linear("linear_hessian_w6a6_tinygrad", shape_x=(4, 13, 48), oc=24, n_V=1, w_bit=6, a_bit=6,
               grad_scale=1e-10, seed=1, **PTQ4VIT)
    gen_linear("linear_cosine_w8a8", shape_x=(4, 13, 48), oc=36, n_V=3, w_bit=8, a_bit=8, seed=2, **BASEPTQ)
    gen_linear("linear_head2d_hessian", shape_x=(8, 40), oc=10, n_V=1, w_bit=8, a_bit=8, seed=3, **PTQ4VIT)
    gen_linear("linear_nobias_L2", shape_x=(3, 7, 32), oc=16, n_V=2, w_bit=8, a_bit=8, seed=4, bias=False,
               metric="L2_norm", eq_alpha=0.3, eq_beta=1.3, eq_n=50, search_round=2)
    gen_linear("linear_blocks_nH2_na2", shape_x=(3, 7, 32), oc=16, n_V=2, n_H=2, n_a=2, w_bit=8, a_bit=8, seed=5,
               metric="hessian", eq_alpha=0.2, eq_beta=1.2, eq_n=40, search_round=2)
    for i, met in enumerate(["L1_norm", "linear_weighted_L2_norm", "square_weighted_L2_norm"]):
        gen_linear(f"linear_metric_{met}", shape_x=(3, 5, 32), oc=16, n_V=1, w_bit=8, a_bit=8, seed=6 + i,
                   metric=met, eq_alpha=0.3, eq_beta=1.2, eq_n=30, search_round=1)
    gen_linear("postgelu_hessian_w8a8", shape_x=(4, 13, 64), oc=24, postgelu=True, n_V=1, w_bit=8, a_bit=8,
               seed=10, **PTQ4VIT)
    gen_linear("postgelu_hessian_w6a6", shape_x=(4, 13, 64), oc=24, postgelu=True, n_V=1, w_bit=6, a_bit=6,
               seed=11, **PTQ4VIT)
    gen_linear("postgelu_cosine_w8a8", shape_x=(4, 13, 64), oc=24, postgelu=True, n_V=1, w_bit=8, a_bit=8,
               seed=12, **BASEPTQ)
    # ---- MatMul (quant_layers/matmul.py:390-644) ---------------------------------
    gen_matmul("matmul_qk_hessian_w8a8", b=4, H=3, d1=13, d2=8, d3=13, A_bit=8, B_bit=8, seed=20, **PTQ4VIT)
    gen_matmul("matmul_qk_cosine_w6a6", b=4, H=3, d1=13, d2=8, d3=13, A_bit=6, B_bit=6, seed=21, **BASEPTQ)
    gen_matmul("matmul_sv_hessian_plain", b=4, H=3, d1=13, d2=13, d3=8, A_bit=8, B_bit=8, seed=22, **PTQ4VIT)
    gen_matmul("matmul_sos_hessian_w8a8", b=4, H=3, d1=13, d2=13, d3=8, sos=True, A_bit=8, B_bit=8, seed=23, **PTQ4VIT)
    gen_matmul("matmul_sos_hessian_w6a6", b=4, H=3, d1=13, d2=13, d3=8, sos=True, A_bit=6, B_bit=6, seed=24, **PTQ4VIT)
    # ---- Conv2d (quant_layers/conv.py:279-614) -----------------------------------
    gen_conv("conv_channelwise_hessian", b=4, ic=3, hw=32, oc=12, k=8, stride=8, w_bit=8, a_bit=32, seed=30, **PTQ4VIT)
    gen_conv("conv_channelwise_cosine", b=4, ic=3, hw=32, oc=12, k=8, stride=8, w_bit=8, a_bit=32, seed=31, **BASEPTQ)
    gen_conv("conv_layerwise_cosine", b=4, ic=3, hw=32, oc=12, k=8, stride=8, channelwise=False, w_bit=8, a_bit=32,
             seed=32, **BASEPTQ)
    # NB: the layer-wise class cannot search activations (conv.py:420 indexes a 4-D tensor with dim 4 ->
    # IndexError), so it is only usable with a_bit=32, as both shipped configs do (configs/BasePTQ.py:50).
    gen_conv("conv_layerwise_hessian_w6", b=4, ic=3, hw=32, oc=12, k=8, stride=8, channelwise=False, w_bit=6, a_bit=32,
             seed=33, **PTQ4VIT)
    gen_ptqsl_convs()
    gen_conv("conv_channelwise_hessian_a8_overlap", b=3, ic=3, hw=20, oc=8, k=5, stride=3, w_bit=8, a_bit=8, seed=34,
             metric="hessian", eq_alpha=0.3, eq_beta=1.2, eq_n=30, search_round=2)




def gen_prune_eligible():
    """Layer cases the reference runs in seconds that are LARGE ENOUGH for the build's exact candidate pruning to engage
    (csrc/p4v_api.hip::run_pass_pruned: a Linear needs >= 640 samples, a MatMul >= 64 query rows) with the gradient profile
    of a ViT (class-token rows carry the weight, |g| ~ 1e-10 as the reference's KL loss produces): the
    reference's own intervals and score tables pin the DEFAULT call of the engine -- no score tables requested, pruned
    passes -- in tests/test_hip_production_path.py.  Prefix `prune_`: the generic parametrised golden tests do not load
    these (they are ~1-2 MB each)."""
    _install_shims()
    os.chdir(REF)
    gen_linear("prune_linear_qkv_hessian_w8a8", shape_x=(4, 197, 128), oc=192, n_V=3, w_bit=8, a_bit=8, seed=70,
               grad_scale=1e-10, cls_heavy=True, store_qf=False, **PTQ4VIT)
    gen_linear("prune_postgelu_hessian_w8a8", shape_x=(4, 197, 256), oc=64, postgelu=True, n_V=1, w_bit=8, a_bit=8, seed=71,
               grad_scale=1e-10, cls_heavy=True, store_qf=False, **PTQ4VIT)
    gen_matmul("prune_matmul_qk_hessian_w8a8", b=2, H=2, d1=197, d2=32, d3=197, A_bit=8, B_bit=8, seed=72,
               grad_scale=1e-10, cls_heavy=True, store_qf=False, **PTQ4VIT)
    gen_matmul("prune_matmul_sos_hessian_w8a8", b=2, H=2, d1=197, d2=197, d3=32, sos=True, A_bit=8, B_bit=8, seed=73,
               grad_scale=1e-10, cls_heavy=True, store_qf=False, **PTQ4VIT)


# --------------------------------------------------------------------------- #
# whole-calibrator fixture: the reference's net_wrap + HessianQuantCalibrator on a mini ViT
# --------------------------------------------------------------------------- #
def gen_mini_vit(name="minivit_ptq4vit"):
    """Reference wrap + batching_quant_calib (reference utils/quant_calib.py:300-378) on a 2-block mini ViT built
    from ptq4vit_amd.utils.models (same forward as the patched timm attention, reference utils/models.py:10-26).
    Stored: images, per-module captured raw_input/raw_out/raw_grad (as the reference's hooks cached them) and the
    calibrated intervals.  The weights are reproducible from the seed."""
    import types
    stub = types.ModuleType("timm")
    for sub in ("timm.models", "timm.models.vision_transformer", "timm.models.swin_transformer"):
        sys.modules[sub] = types.ModuleType(sub)
    sys.modules["timm"] = stub
    stub.models = sys.modules["timm.models"]
    sys.modules["timm.models"].vision_transformer = sys.modules["timm.models.vision_transformer"]
    sys.modules["timm.models.vision_transformer"].Attention = type("Attention", (torch.nn.Module,), {})
    sys.modules["timm.models.swin_transformer"].WindowAttention = type("WindowAttention", (torch.nn.Module,), {})
    import importlib
    ref_models = importlib.import_module("utils.models")
    ref_wrap = importlib.import_module("utils.net_wrap")
    ref_calib = importlib.import_module("utils.quant_calib")
    cfg = importlib.import_module("configs.PTQ4ViT")
    repo = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    if repo not in sys.path:
        sys.path.append(repo)
    from ptq4vit_amd.utils import models as my_models

    kw = dict(img_size=32, patch_size=8, embed_dim=48, depth=2, num_heads=3, num_classes=10)
    net = my_models.get_net("vit_tiny_patch16_224", seed=0, device="cpu", **kw)
    for m in list(net.modules()):           # let the reference's isinstance(m, MatMul) recognise the matmul modules
        for cname, child in list(m.named_children()):
            if isinstance(child, my_models.MatMul):
                setattr(m, cname, ref_models.MatMul())
    wrapped = ref_wrap.wrap_modules_in_net(net, cfg)
    g = torch.Generator().manual_seed(0)
    images = torch.randn(8, 3, 32, 32, generator=g)

    class Loader:
        batch_size = 8

        def __iter__(self):
            yield images, torch.zeros(8, dtype=torch.long)

    captured = {}
    for n, m in wrapped.items():
        orig = m.calibration_step2

        def rec(_orig=orig, _m=m, _n=n):
            ri = _m.raw_input
            captured[_n] = dict(
                raw_input=[t.clone().numpy() for t in ri] if isinstance(ri, (list, tuple)) else ri.clone().numpy(),
                raw_out=_m.raw_out.clone().numpy(), raw_grad=_m.raw_grad.clone().numpy())
            return _orig()
        m.calibration_step2 = rec
    cal = ref_calib.HessianQuantCalibrator(net, wrapped, Loader(), sequential=False, batch_size=4)
    cal.batching_quant_calib()
    with torch.no_grad():
        logits = net(images)
    payload = {"images": images.numpy(), "quant_logits": logits.numpy(), "names": np.array(list(wrapped))}
    for n, m in wrapped.items():
        key = n.replace(".", "__")
        cap = captured[n]
        if isinstance(cap["raw_input"], list):
            payload[f"{key}::A"], payload[f"{key}::B"] = cap["raw_input"]
        else:
            payload[f"{key}::x"] = cap["raw_input"]
        payload[f"{key}::out"], payload[f"{key}::grad"] = cap["raw_out"], cap["raw_grad"]
        for a in ("w_interval", "a_interval", "A_interval", "B_interval", "split"):
            v = getattr(m, a, None)
            if v is not None:
                payload[f"{key}::{a}"] = np.asarray(v)
    payload["model_kwargs"] = np.array(json.dumps(kw))
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **payload)
    print(f"wrote {name}.npz ({len(wrapped)} modules)")


def _reference_harness(cfg_name):
    """Import the reference's wrap / calibrator / config modules with timm stubbed."""
    import types, importlib
    _install_shims()
    os.chdir(REF)
    for sub in ("timm", "timm.models", "timm.models.vision_transformer", "timm.models.swin_transformer"):
        sys.modules.setdefault(sub, types.ModuleType(sub))
    sys.modules["timm.models.vision_transformer"].Attention = type("Attention", (torch.nn.Module,), {})
    sys.modules["timm.models.swin_transformer"].WindowAttention = type("WindowAttention", (torch.nn.Module,), {})
    ref_models = importlib.import_module("utils.models")
    ref_wrap = importlib.import_module("utils.net_wrap")
    ref_calib = importlib.import_module("utils.quant_calib")
    cfg = importlib.import_module("configs." + cfg_name)
    repo = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    if repo not in sys.path:
        sys.path.append(repo)
    from ptq4vit_amd.utils import models as my_models
    return ref_models, ref_wrap, ref_calib, cfg, my_models


def _reference_net(my_models, ref_models, name, **kw):
    net = my_models.get_net(name, seed=0, device="cpu", **kw)
    for m in list(net.modules()):           # let the reference's isinstance(m, MatMul) recognise the matmul modules
        for cname, child in list(m.named_children()):
            if isinstance(child, my_models.MatMul):
                setattr(m, cname, ref_models.MatMul())
    return net


class _Loader:
    def __init__(self, images):
        self.images, self.batch_size = images, images.shape[0]

    def __iter__(self):
        yield self.images, torch.zeros(self.images.shape[0], dtype=torch.long)


def _interval_payload(wrapped):
    payload = {"names": np.array(list(wrapped))}
    for n, m in wrapped.items():
        key = n.replace(".", "__")
        for a in ("w_interval", "a_interval", "A_interval", "B_interval", "split"):
            v = getattr(m, a, None)
            if v is None:
                continue
            if isinstance(v, (list, tuple)):      # non-batching post-GELU class: [positive (n_a, 1), negative scalar]
                payload[f"{key}::{a}"] = np.asarray(v[0])
                payload[f"{key}::a_neg_interval"] = np.asarray(v[1], dtype=np.float64)
            else:
                payload[f"{key}::{a}"] = np.asarray(v)
    return payload


def gen_deit_tiny(name="deit_tiny_224_baseptq_4img"):
    """BASELINE.json config 0 run by the REFERENCE itself: DeiT-tiny/224 (the build's restatement of the architecture,
    seeded random weights -- timm / checkpoints are not available offline), configs/BasePTQ.py as shipped (cosine metric,
    one round; cosine does not read raw_grad), 4 seeded calibration images, reference utils/net_wrap.py:39-81 +
    HessianQuantCalibrator.batching_quant_calib (utils/quant_calib.py:300-378) on the CPU.  Stored: every module's
    calibrated intervals, the raw and the quantised logits of the calibration images.  Weights and images are
    reproducible from the seeds (checksums stored), so the fixture is KBs."""
    ref_models, ref_wrap, ref_calib, cfg, my_models = _reference_harness("BasePTQ")
    net = _reference_net(my_models, ref_models, "deit_tiny_patch16_224")
    g = torch.Generator().manual_seed(0)
    images = torch.randn(4, 3, 224, 224, generator=g)
    with torch.no_grad():
        raw_logits = net(images)
    wrapped = ref_wrap.wrap_modules_in_net(net, cfg)
    tables = {}
    for n, m in wrapped.items():            # every score table the reference feeds to argmax, per module, in call order
        orig = m.calibration_step2

        def rec(_orig=orig, _n=n):
            with ArgmaxRecorder() as r:
                out = _orig()
            tables[_n] = r.tables
            return out
        m.calibration_step2 = rec
    cal = ref_calib.HessianQuantCalibrator(net, wrapped, _Loader(images), sequential=False, batch_size=4)
    cal.batching_quant_calib()
    with torch.no_grad():
        logits = net(images)
    payload = _interval_payload(wrapped)
    for n, tabs in tables.items():
        for i, t in enumerate(tabs):
            payload[f"{n.replace('.', '__')}::scores_{i}"] = t
    payload.update(raw_logits=raw_logits.numpy(), quant_logits=logits.numpy(),
                   images_sum=np.array(images.double().sum().item()), images_abs_sum=np.array(images.double().abs().sum().item()),
                   weights_abs_sum=np.array(sum(p.double().abs().sum().item() for p in net.parameters())),
                   config=np.array(json.dumps(dict(model="deit_tiny_patch16_224", cfg="BasePTQ", images=4, image_seed=0, net_seed=0,
                                                   batch_size=4, sequential=False))))
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **payload)
    print(f"wrote {name}.npz ({len(wrapped)} modules)")


def gen_deit_tiny_eval(name="deit_tiny_224_baseptq_eval1000", n_eval=1000, eval_seed=7):
    """The evaluation loop of the reference (example/test_vit.py:26-45: argmax of the quantised network's logits per image) on
    the network of `gen_deit_tiny` -- DeiT-tiny/224, BasePTQ, calibrated BY THE REFERENCE on the 4 seeded images -- over 1000
    seeded evaluation images: the reference's top-1 prediction of every image, its top-1 / top-2 logit margin, and the raw
    network's prediction (the "label" a data-free top-1 uses).  ImageNet does not exist offline; this pins the inference path
    and the top-1 bookkeeping on what does.  The calibration must reproduce the intervals of deit_tiny_224_baseptq_4img.npz."""
    ref_models, ref_wrap, ref_calib, cfg, my_models = _reference_harness("BasePTQ")
    net = _reference_net(my_models, ref_models, "deit_tiny_patch16_224")
    images = torch.randn(4, 3, 224, 224, generator=torch.Generator().manual_seed(0))
    ev = torch.randn(n_eval, 3, 224, 224, generator=torch.Generator().manual_seed(eval_seed))
    with torch.no_grad():
        raw = torch.cat([net(ev[i:i + 50]) for i in range(0, n_eval, 50)])
    wrapped = ref_wrap.wrap_modules_in_net(net, cfg)
    ref_calib.HessianQuantCalibrator(net, wrapped, _Loader(images), sequential=False, batch_size=4).batching_quant_calib()
    stored = np.load(os.path.join(OUT, "deit_tiny_224_baseptq_4img.npz"), allow_pickle=False)
    for k, v in _interval_payload(wrapped).items():
        if "::" in k:
            assert np.array_equal(np.asarray(v), stored[k]), f"{k}: not the calibration of deit_tiny_224_baseptq_4img.npz"
    with torch.no_grad():
        q = torch.cat([net(ev[i:i + 50]) for i in range(0, n_eval, 50)])
    top2 = q.topk(2, dim=1).values
    rng = float(q.max() - q.min())
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"),
                        quant_argmax=q.argmax(1).numpy().astype(np.int16), quant_margin=(top2[:, 0] - top2[:, 1]).numpy().astype(np.float32),
                        raw_argmax=raw.argmax(1).numpy().astype(np.int16), quant_logits_head=q[:16].numpy(), logit_range=np.array(rng),
                        eval_sum=np.array(ev.double().sum().item()), eval_abs_sum=np.array(ev.double().abs().sum().item()),
                        config=np.array(json.dumps(dict(model="deit_tiny_patch16_224", cfg="BasePTQ", calib_images=4, image_seed=0,
                                                        net_seed=0, eval_images=n_eval, eval_seed=eval_seed))))
    agree = int((q.argmax(1) == raw.argmax(1)).sum())
    print(f"wrote {name}.npz: reference top-1 == raw network's top-1 on {agree}/{n_eval} images, logit range {rng:.3f}")


# --------------------------------------------------------------------------- #
# the non-batching classes and the other calibrator entry points
# --------------------------------------------------------------------------- #
def gen_ptqsl_linear(name, *, shape_x, oc, postgelu=False, grad_scale=1e-3, seed=0, bias=True, **kw):
    """PTQSLQuantLinear / PostGeluPTQSLQuantLinear.calibration_step2(x) (linear.py:94-347): raw_out / raw_grad cached,
    the input handed over as the argument; scores are means over (batch, tokens) instead of the batching classes' sums."""
    from quant_layers.linear import PTQSLQuantLinear, PostGeluPTQSLQuantLinear

    g = torch.Generator().manual_seed(seed)
    ic = shape_x[-1]
    w = torch.randn(oc, ic, generator=g) * 0.05 * torch.linspace(0.5, 2.0, oc).view(-1, 1)
    b = torch.randn(oc, generator=g) * 0.1 if bias else None
    x = torch.randn(*shape_x, generator=g)
    if postgelu:
        x = F.gelu(1.5 * x)
    out = F.linear(x, w, b)
    grad = torch.randn(out.shape, generator=g) * grad_scale
    m = (PostGeluPTQSLQuantLinear if postgelu else PTQSLQuantLinear)(ic, oc, bias=bias, **kw)
    m.weight.data = w.clone()
    if bias:
        m.bias.data = b.clone()
    m.raw_input, m.raw_out = x.clone(), out.clone()
    m.raw_grad = grad.clone() if kw.get("metric") == "hessian" else None
    with torch.no_grad(), ArgmaxRecorder() as rec:
        qf = m.calibration_step2(x.clone())
    a_iv = m.a_interval[0] if postgelu else m.a_interval
    arrays = dict(weight=w.numpy(), x=x.numpy(), out=out.numpy(), grad=grad.numpy(), w_interval=m.w_interval.numpy(),
                  a_interval=np.asarray(a_iv), quant_forward=qf.numpy())
    if bias:
        arrays["bias"] = b.numpy()
    _save(name, dict(kind="ptqsl_linear", postgelu=postgelu, oc=oc, **kw), arrays, rec.tables)


def gen_ptqsl_matmul(name, *, b, H, d1, d2, d3, sos=False, grad_scale=1e-3, seed=0, **kw):
    """PTQSLQuantMatMul / SoSPTQSLQuantMatMul.calibration_step2(A, B) (matmul.py:62-388): n_G as configured (NOT forced to
    the head count as in the batching classes), group scores = mean over the heads of a group."""
    from quant_layers.matmul import PTQSLQuantMatMul, SoSPTQSLQuantMatMul

    g = torch.Generator().manual_seed(seed)
    if sos:
        A = torch.softmax(torch.randn(b, H, d1, d2, generator=g) * 3.0, dim=-1)
    else:
        A = torch.randn(b, H, d1, d2, generator=g) * torch.linspace(0.5, 2.0, H).view(1, H, 1, 1)
    Bm = (torch.randn(b, H, d3, d2, generator=g) * torch.linspace(2.0, 0.5, H).view(1, H, 1, 1)).transpose(-2, -1)
    out = A @ Bm
    grad = torch.randn(out.shape, generator=g) * grad_scale
    m = (SoSPTQSLQuantMatMul if sos else PTQSLQuantMatMul)(**kw)
    m.raw_input, m.raw_out = [A.clone(), Bm.clone()], out.clone()
    m.raw_grad = grad.clone() if kw.get("metric") == "hessian" else None
    with torch.no_grad(), ArgmaxRecorder() as rec:
        qf = m.calibration_step2(A.clone(), Bm.clone())
    arrays = dict(A=A.numpy(), B=Bm.contiguous().numpy(), out=out.numpy(), grad=grad.numpy(),
                  A_interval=np.asarray(m.A_interval), B_interval=m.B_interval.numpy(), quant_forward=qf.numpy())
    if sos:
        arrays["split"] = np.asarray(m.split)
    _save(name, dict(kind="ptqsl_matmul", sos=sos, **kw), arrays, rec.tables)


def gen_quantile_conv(name="quantileconv_w8a8", *, b=3, ic=3, hw=24, oc=8, k=4, stride=4, seed=60, **kw):
    """QuantileQuantConv2d.calibration_step2(x) (conv.py:91-124): intervals from the 0.9999 quantiles."""
    from quant_layers.conv import QuantileQuantConv2d

    g = torch.Generator().manual_seed(seed)
    w = torch.randn(oc, ic, k, k, generator=g) * 0.05
    bias = torch.randn(oc, generator=g) * 0.1
    x = torch.randn(b, ic, hw, hw, generator=g)
    m = QuantileQuantConv2d(ic, oc, k, stride, **kw)
    m.weight.data = w.clone()
    m.bias.data = bias.clone()
    with torch.no_grad():
        qf = m.calibration_step2(x.clone())
    _save(name, dict(kind="quantile_conv", stride=stride, **kw),
          dict(weight=w.numpy(), bias=bias.numpy(), x=x.numpy(), w_interval=np.asarray(m.w_interval),
               a_interval=np.asarray(m.a_interval), quant_forward=qf.numpy()), [])


def gen_f4_layers():
    hs = dict(metric="hessian", eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=2)
    gen_ptqsl_linear("ptqsllinear_hessian_v3", shape_x=(4, 13, 48), oc=36, n_V=3, w_bit=8, a_bit=8, seed=50, **hs)
    gen_ptqsl_linear("ptqsllinear_cosine_head2d", shape_x=(8, 40), oc=10, n_V=1, w_bit=8, a_bit=8, seed=51,
                     metric="cosine", eq_alpha=0.5, eq_beta=1.2, eq_n=100, search_round=1)
    gen_ptqsl_linear("ptqsllinear_l2_blocks_h2a2_w6", shape_x=(3, 7, 32), oc=16, n_V=2, n_H=2, n_a=2, w_bit=6, a_bit=6, seed=52,
                     metric="L2_norm", eq_alpha=0.2, eq_beta=1.2, eq_n=40, search_round=2)
    gen_ptqsl_linear("ptqslpostgelu_hessian_w8a8", shape_x=(4, 13, 64), oc=24, postgelu=True, n_V=1, w_bit=8, a_bit=8,
                     seed=53, **hs)
    gen_ptqsl_linear("ptqslpostgelu_l2_w6a6", shape_x=(4, 13, 64), oc=24, postgelu=True, n_V=2, w_bit=6, a_bit=6, seed=54,
                     metric="L2_norm", eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=2)
    gen_ptqsl_matmul("ptqslmatmul_qk_hessian_g3", b=4, H=3, d1=13, d2=8, d3=13, A_bit=8, B_bit=8, seed=55,
                     n_G_A=3, n_G_B=3, **hs)
    gen_ptqsl_matmul("ptqslmatmul_qk_l2_g1", b=4, H=3, d1=13, d2=8, d3=13, A_bit=8, B_bit=8, seed=56, n_G_A=1, n_G_B=1,
                     metric="L2_norm", eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=2)
    gen_ptqsl_matmul("ptqslmatmul_sv_hessian_g2of5_w6", b=3, H=5, d1=11, d2=11, d3=8, A_bit=6, B_bit=6, seed=57,
                     n_G_A=2, n_G_B=2, **hs)                       # 5 heads in 2 groups: crb_groups 3, one padding head
    gen_ptqsl_matmul("ptqslsos_hessian_g3", b=4, H=3, d1=13, d2=13, d3=8, sos=True, A_bit=8, B_bit=8, seed=58, n_G_B=3, **hs)
    gen_ptqsl_matmul("ptqslsos_l2_g1_w6", b=4, H=3, d1=13, d2=13, d3=8, sos=True, A_bit=6, B_bit=6, seed=59, n_G_B=1,
                     metric="L2_norm", eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=2)
    gen_quantile_conv()


def gen_f4_calibrators():
    """The reference's other calibrator entry points on the 2-block mini ViT, every module a NON-batching class (what
    these entry points call: calibration_step1(x) / calibration_step2(x)):
      QuantCalibrator(sequential=True).quant_calib()   -> sequential_quant_calib   (quant_calib.py:28-55)
      QuantCalibrator(sequential=False).quant_calib()  -> parallel_quant_calib     (quant_calib.py:57-93)
      HessianQuantCalibrator(sequential=False, batch_size=4).quant_calib()          (quant_calib.py:216-298)
    Stored per run: the images, every module's intervals, the quantised logits.  Weights come from the seed."""
    ref_models, ref_wrap, ref_calib, _, my_models = _reference_harness("PTQ4ViT")
    from quant_layers.conv import PTQSLQuantConv2d
    from quant_layers.linear import PTQSLQuantLinear, PostGeluPTQSLQuantLinear
    from quant_layers.matmul import PTQSLQuantMatMul, SoSPTQSLQuantMatMul
    kw = dict(img_size=32, patch_size=8, embed_dim=48, depth=2, num_heads=3, num_classes=10)
    g = torch.Generator().manual_seed(3)
    images = torch.randn(8, 3, 32, 32, generator=g)
    for run, metric in (("sequential", "L2_norm"), ("parallel", "L2_norm"), ("hessian", "hessian")):
        hp = dict(metric=metric, search_round=2, eq_alpha=0.01, eq_beta=1.2, eq_n=100)

        class cfg:  # noqa: N801  (the factory protocol of configs/PTQ4ViT.py:51-80 over the non-batching classes)
            @staticmethod
            def get_module(kind, *a, **k):
                if kind == "qconv":
                    return PTQSLQuantConv2d(*a, **k, w_bit=8, a_bit=8, n_V=1, n_H=1, **hp)
                if kind == "qlinear_MLP_2":
                    return PostGeluPTQSLQuantLinear(*a, **k, w_bit=8, a_bit=8, **hp)
                if kind.startswith("qlinear"):
                    return PTQSLQuantLinear(*a, **k, w_bit=8, a_bit=8, n_V=3 if kind == "qlinear_qkv" else 1, **hp)
                if kind == "qmatmul_scorev":
                    return SoSPTQSLQuantMatMul(A_bit=8, B_bit=8, n_G_B=3, **hp)
                return PTQSLQuantMatMul(A_bit=8, B_bit=8, n_G_A=3, n_G_B=3, **hp)

        net = _reference_net(my_models, ref_models, "vit_tiny_patch16_224", **kw)
        wrapped = ref_wrap.wrap_modules_in_net(net, cfg)
        if run == "hessian":
            ref_calib.HessianQuantCalibrator(net, wrapped, _Loader(images), sequential=False, batch_size=4).quant_calib()
        else:
            ref_calib.QuantCalibrator(net, wrapped, _Loader(images), sequential=(run == "sequential")).quant_calib()
        with torch.no_grad():
            logits = net(images)
        payload = _interval_payload(wrapped)
        payload.update(images=images.numpy(), quant_logits=logits.numpy(), model_kwargs=np.array(json.dumps(kw)),
                       hp=np.array(json.dumps(hp)))
        np.savez_compressed(os.path.join(OUT, f"minivit_calibrator_{run}.npz"), **payload)
        print(f"wrote minivit_calibrator_{run}.npz ({len(wrapped)} modules)")


def gen_integer(name="minivit_integer"):
    """Reference utils/integer.py (quantize_int_weight, quantize_int_activation, get_model_int_weight) applied to
    the calibrated mini ViT of minivit_ptq4vit.npz: the reference modules get the stored intervals, the stored
    captured inputs are pushed through the reference's pre-hook.  Stored: int8 / uint8 images only."""
    import types, importlib
    _install_shims()
    os.chdir(REF)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for sub in ("timm", "timm.models", "timm.models.vision_transformer", "timm.models.swin_transformer"):
        sys.modules.setdefault(sub, types.ModuleType(sub))
    sys.modules["timm.models.vision_transformer"].Attention = type("Attention", (torch.nn.Module,), {})
    sys.modules["timm.models.swin_transformer"].WindowAttention = type("WindowAttention", (torch.nn.Module,), {})
    ref_models = importlib.import_module("utils.models")
    ref_wrap = importlib.import_module("utils.net_wrap")
    ref_int = importlib.import_module("utils.integer")
    cfg = importlib.import_module("configs.PTQ4ViT")
    repo = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    if repo not in sys.path:
        sys.path.append(repo)
    from ptq4vit_amd.utils import models as my_models
    g = np.load(os.path.join(OUT, "minivit_ptq4vit.npz"), allow_pickle=False)
    kw = json.loads(str(g["model_kwargs"]))
    net = my_models.get_net("vit_tiny_patch16_224", seed=0, device="cpu", **kw)
    for m in list(net.modules()):
        for cname, child in list(m.named_children()):
            if isinstance(child, my_models.MatMul):
                setattr(m, cname, ref_models.MatMul())
    wrapped = ref_wrap.wrap_modules_in_net(net, cfg)
    payload = {}
    for n, m in wrapped.items():
        key = n.replace(".", "__")
        for a in ("w_interval", "a_interval", "A_interval", "B_interval", "split"):
            if f"{key}::{a}" in g.files:
                setattr(m, a, torch.from_numpy(g[f"{key}::{a}"]))
        m.calibrated = True
        inputs = (torch.from_numpy(g[f"{key}::A"]), torch.from_numpy(g[f"{key}::B"])) if f"{key}::A" in g.files \
            else (torch.from_numpy(g[f"{key}::x"]),)
        if hasattr(m, "a_bit") and m.a_bit != 8:     # the patch-embedding conv keeps fp32 activations (PTQ4ViT.py:54)
            continue
        if len(inputs) == 2:
            m._get_padding_parameters(*inputs)          # crb_* / pad_* (set by calibration in a real run, matmul.py:411-417)
        ref_int.quantize_int_activation(m, inputs)
        for i, t in enumerate(getattr(m, "int_input", [])):
            payload[f"{key}::int_input{i}"] = t.numpy()
    for n, w_int in ref_int.get_model_int_weight(wrapped).items():
        payload[f"{n.replace('.', '__')}::w_int"] = w_int.numpy()
        m = wrapped[n]
        payload[f"{n.replace('.', '__')}::w_deq"] = ref_int.dequantize_int_weight(m, w_int).numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **payload)
    print(f"wrote {name}.npz ({len(payload)} arrays)")


def gen_swin_attention(name="swin_window_attention"):
    """Reference utils/models.py:28-56 `window_attention_forward` bound onto a window-attention module carrying the
    build's parameters (timm's WindowAttention attributes: qkv, proj, relative_position_bias_table / _index, softmax,
    scale, window_size, num_heads + the two MatMul modules): input, shift mask, output."""
    import types, importlib
    from types import MethodType
    _install_shims()
    os.chdir(REF)
    for sub in ("timm", "timm.models", "timm.models.vision_transformer", "timm.models.swin_transformer"):
        sys.modules.setdefault(sub, types.ModuleType(sub))
    sys.modules["timm.models.vision_transformer"].Attention = type("Attention", (torch.nn.Module,), {})
    sys.modules["timm.models.swin_transformer"].WindowAttention = type("WindowAttention", (torch.nn.Module,), {})
    ref_models = importlib.import_module("utils.models")
    repo = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    if repo not in sys.path:
        sys.path.append(repo)
    from ptq4vit_amd.utils import models as my_models
    torch.manual_seed(0)
    blk = my_models.SwinBlock(24, (14, 14), num_heads=3, window_size=7, shift_size=3)
    att = blk.attn
    torch.nn.init.trunc_normal_(att.relative_position_bias_table, std=0.5)
    att.matmul1, att.matmul2 = ref_models.MatMul(), ref_models.MatMul()
    att.forward = MethodType(ref_models.window_attention_forward, att)
    x = torch.randn(8, 49, 24)          # 2 images x 4 windows
    with torch.no_grad():
        y_mask = att(x, blk.attn_mask)
        y_nomask = att(x)
    sd = {k: v.numpy() for k, v in att.state_dict().items()}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), x=x.numpy(), mask=blk.attn_mask.numpy(), y_mask=y_mask.numpy(),
                        y_nomask=y_nomask.numpy(), **{"sd::" + k: v for k, v in sd.items()})
    print(f"wrote {name}.npz")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "swin":
        gen_swin_attention()
    elif len(sys.argv) > 1 and sys.argv[1] == "deit":
        gen_deit_tiny()
    elif len(sys.argv) > 1 and sys.argv[1] == "deit_eval":
        gen_deit_tiny_eval()
    elif len(sys.argv) > 1 and sys.argv[1] == "f4":
        _install_shims()
        os.chdir(REF)
        gen_f4_layers()
        gen_f4_calibrators()
    elif len(sys.argv) > 1 and sys.argv[1] == "integer":
        gen_integer()
    elif len(sys.argv) > 1 and sys.argv[1] == "prune":
        gen_prune_eligible()
    else:
        main(sys.argv[1] if len(sys.argv) > 1 else None)
        if len(sys.argv) == 1:
            gen_mini_vit()
            gen_integer()
            gen_swin_attention()
            gen_deit_tiny()
            gen_deit_tiny_eval()
            gen_f4_layers()
            gen_f4_calibrators()
            gen_prune_eligible()
