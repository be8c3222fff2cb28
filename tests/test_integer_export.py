"""utils/integer.py against the fixture made by running the reference's utils/integer.py on
the calibrated mini ViT (oracle/gen_golden.py integer).  Integer formats: bit-exact."""
import json

import numpy as np
import torch

from ptq4vit_amd.configs import PTQ4ViT
from ptq4vit_amd.quant_layers.matmul import MinMaxQuantMatMul
from ptq4vit_amd.utils import integer, models, net_wrap


def _calibrated_mini():
    g = np.load("tests/golden/minivit_ptq4vit.npz", allow_pickle=False)
    kw = json.loads(str(g["model_kwargs"]))
    net = models.get_net("vit_tiny_patch16_224", seed=0, device="cpu", **kw)
    wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
    for n, m in wrapped.items():
        key = n.replace(".", "__")
        for a in ("w_interval", "a_interval", "A_interval", "B_interval", "split"):
            if f"{key}::{a}" in g.files:
                setattr(m, a, torch.from_numpy(g[f"{key}::{a}"]))
        m.calibrated = True
    return g, wrapped


def test_int_weights_match_reference():
    g, wrapped = _calibrated_mini()
    gi = np.load("tests/golden/minivit_integer.npz", allow_pickle=False)
    ws = integer.get_model_int_weight(wrapped)
    expect = {k[:-len("::w_int")].replace("__", ".") for k in gi.files if k.endswith("::w_int")}
    assert set(ws) == expect and len(ws) > 0
    for n, w_int in ws.items():
        key = n.replace(".", "__")
        assert w_int.dtype == torch.int8
        np.testing.assert_array_equal(w_int.numpy(), gi[f"{key}::w_int"])
        np.testing.assert_array_equal(integer.dequantize_int_weight(wrapped[n], w_int).numpy(), gi[f"{key}::w_deq"])


def test_int_activations_match_reference():
    g, wrapped = _calibrated_mini()
    gi = np.load("tests/golden/minivit_integer.npz", allow_pickle=False)
    checked = 0
    for n, m in wrapped.items():
        key = n.replace(".", "__")
        if f"{key}::int_input0" not in gi.files:
            continue
        if isinstance(m, MinMaxQuantMatMul):
            inputs = (torch.from_numpy(g[f"{key}::A"]), torch.from_numpy(g[f"{key}::B"]))
            m._get_padding_parameters(*inputs)
        else:
            inputs = (torch.from_numpy(g[f"{key}::x"]),)
        integer.quantize_int_activation(m, inputs)
        for i, t in enumerate(m.int_input):
            ref = gi[f"{key}::int_input{i}"]
            assert str(t.dtype).replace("torch.", "") == str(ref.dtype), (n, i, t.dtype, ref.dtype)
            np.testing.assert_array_equal(t.numpy(), ref)
            checked += 1
    assert checked >= 14      # 2 blocks x (qkv, proj, fc1, fc2 + 2 matmuls x 2 operands) + head


def test_twin_formats_dtypes_and_wraparound():
    """Twin operands are stored as uint8 sums `(region + 128) + other-region` exactly as the reference computes them
    (integer.py:63-71,88-96) -- including its uint8 wrap-around for softmax values far above the split."""
    g, wrapped = _calibrated_mini()
    fc2 = wrapped["blocks.0.mlp.fc2"]
    x = torch.from_numpy(g["blocks__0__mlp__fc2::x"])
    integer.quantize_int_activation(fc2, (x,))
    q = fc2.int_input[0]
    assert q.dtype == torch.uint8 and bool((q >= 128).all())
    k_pos = torch.clamp(torch.round(x / fc2.a_interval), 0, 127)
    k_neg = torch.clamp(torch.round(x / fc2.a_neg_interval), -127, 0).abs()
    assert torch.equal(q.to(torch.int32), (128 + k_pos + k_neg).to(torch.int32))
    sv = wrapped["blocks.0.attn.matmul2"]
    A, B = torch.from_numpy(g["blocks__0__attn__matmul2::A"]), torch.from_numpy(g["blocks__0__attn__matmul2::B"])
    sv._get_padding_parameters(A, B)
    integer.quantize_int_activation(sv, (A, B))
    qa, qb = sv.int_input
    assert qa.dtype == torch.uint8 and qb.dtype == torch.int8
    hi = torch.clamp(torch.round(A.clamp(sv.split, 1) * 127), 0, 127)
    lo = torch.clamp(torch.round(A.clamp(0, sv.split) / sv.A_interval), 0, 127)
    assert torch.equal(qa.to(torch.int32), ((128 + hi + lo).to(torch.int32)) % 256)


def test_half_precision_non_dense_views_are_read_through_their_own_strides():
    """A sliced / expanded fp16 operand: the export must read the elements the VIEW addresses (its strides were taken
    before the dtype conversion; `.float()` of such a view is dense and has other strides)."""
    g, wrapped = _calibrated_mini()
    fc1 = wrapped["blocks.0.mlp.fc1"]
    x = torch.from_numpy(g["blocks__0__mlp__fc1::x"])
    big = torch.zeros(x.shape[0], x.shape[1], 2 * x.shape[2], dtype=torch.float16)
    big[..., ::2] = x.half()
    view = big[..., ::2]                                   # stride 2 in the last dimension, fp16
    assert not view.is_contiguous()
    integer.quantize_int_activation(fc1, (view,))
    got = fc1.int_input[0]
    integer.quantize_int_activation(fc1, (x.half().float(),))
    assert torch.equal(got, fc1.int_input[0])
    row = x[:1, :1].half().expand(x.shape[0], x.shape[1], x.shape[2])      # stride-0 (expanded) fp16 view
    integer.quantize_int_activation(fc1, (row,))
    got = fc1.int_input[0]
    integer.quantize_int_activation(fc1, (row.float().contiguous(),))
    assert torch.equal(got, fc1.int_input[0])


def test_half_precision_slice_of_a_large_buffer_converts_only_its_span():
    """A small fp16 view into a big buffer (an activation slice): the float32 image has the view's shape / strides / values and
    its storage is the view's span, not the buffer (the whole-storage conversion allocated 4 bytes per buffer element)."""
    big = torch.randn(1 << 20, dtype=torch.float32).half()
    view = big[4096:4096 + 6 * 40].view(6, 40)[1:5, ::2]              # offset, gapped columns: strides (40, 2)
    img = integer._f32_same_strides(view)
    assert img.dtype == torch.float32 and tuple(img.shape) == (4, 20) and tuple(img.stride()) == (40, 2)
    assert torch.equal(img, view.float())
    assert img.untyped_storage().nbytes() <= 4 * (1 + 3 * 40 + 19 * 2)
    assert integer._f32_same_strides(big[:0]).numel() == 0
    f32 = torch.randn(8, 8)[:, ::2]
    assert integer._f32_same_strides(f32).data_ptr() == f32.data_ptr()      # float32 passes through untouched
