"""GPU, through the C ABI: the integer side of the path is BIT-EXACT.

* every operand plane the candidate sweeps consume (p4v_pack_plane_i8 = the search's pack kernel, one candidate):
  symmetric, post-GELU twin positive / negative range (linear.py:601-607), split-of-softmax high / low range
  (matmul.py:595-598) with planted boundary cases, 8 and 6 bit;
* p4v_fake_quant;
* the integer export formats (utils/integer.py -> p4v_export_quantize) against the fixture produced by the
  reference's own utils/integer.py;
* quant_forward of calibrated modules (p4v_linear_quant_forward / p4v_matmul_quant_forward, int8 MFMA) against the
  `quant_forward` arrays the REFERENCE produced for every golden layer case.
"""
import json

import numpy as np
import pytest
import torch

from tests.helpers import golden_names, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from ptq4vit_amd import engine
    return engine


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ---- operand planes ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("bit", [8, 6])
@pytest.mark.parametrize("cols", [199, 256, 16, 7])      # ragged tails, whole 16-element runs, a single run, shorter than a run
def test_twin_postgelu_planes_bit_exact(eng, bit, cols):
    """k_pos = clamp(rint(x/s_pos), 0, q-1), k_neg = clamp(rint(x/s_neg), -q, 0) (linear.py:605-606) -- the two planes of
    the twin weight search; exact halves and range ends planted."""
    from oracle.ptq4vit_oracle import POSTGELU_NEG_RANGE, twin_planes
    q = 2 ** (bit - 1)
    rng = np.random.default_rng(bit * 1000 + cols)
    x = torch.nn.functional.gelu(torch.from_numpy(rng.standard_normal((131, cols)).astype(np.float32) * 1.5)).numpy()
    s_pos = np.float32(x.max() / (q - 0.5) * 0.37)
    s_neg = np.float32(POSTGELU_NEG_RANGE / q)
    n = min(cols, 40)
    x[0, :n] = (np.arange(n, dtype=np.float32) + 0.5) * s_pos              # exact .5 cases of the positive grid
    x[1, :n] = -(np.arange(n, dtype=np.float32) + 0.5) * s_neg             # ... of the negative grid
    x[2, :min(cols, 4)] = np.array([0.0, -0.0, (q - 1) * s_pos * 3, -q * s_neg * 3], np.float32)[:min(cols, 4)]   # zeros, both saturations
    ref_pos, ref_neg = twin_planes(x, s_pos, s_neg, q)
    pos, _ = eng.pack_plane_i8(_t(x), mode="sym", scales=torch.tensor([s_pos]), rows_per_scale=10 ** 9, lo=0, hi=q - 1, qmax=q)
    neg, padded = eng.pack_plane_i8(_t(x), mode="sym", scales=None, const_scale=float(s_neg), lo=-q, hi=0, qmax=q)
    np.testing.assert_array_equal(pos.cpu().numpy(), ref_pos)
    np.testing.assert_array_equal(neg.cpu().numpy(), ref_neg)
    # the merged plane of the large-K weight search (k_sweep7 splits it by sign): k_pos + k_neg, byte for byte
    both, _ = eng.pack_plane_i8(_t(x), mode="twin", scales=torch.tensor([s_pos]), const_scale=float(s_neg), lo=-q, hi=q - 1, qmax=q)
    np.testing.assert_array_equal(both.cpu().numpy(), ref_pos + ref_neg)
    np.testing.assert_array_equal(np.maximum(both.cpu().numpy(), 0), ref_pos)
    np.testing.assert_array_equal(np.minimum(both.cpu().numpy(), 0), ref_neg)
    assert (ref_pos != 0).any() and (ref_neg != 0).any()
    assert not np.logical_and(ref_pos != 0, ref_neg != 0).any()            # disjoint supports
    assert int(padded[:, cols:].abs().max().item() if padded.shape[1] > cols else 0) == 0   # K padding is zero


@pytest.mark.parametrize("bit", [8, 6])
@pytest.mark.parametrize("split_exp", [0, 1, 5, 8, 12, 19])
def test_split_of_softmax_planes_bit_exact(eng, bit, split_exp):
    """matmul.py:596-597 with the non-disjoint ranges of SURVEY.md App. A-9: A == split, A just below / above it,
    A = 0, A = 1, and the split candidates for which round(split (q-1)) is 0 (split (q-1) < 0.5)."""
    from oracle.ptq4vit_oracle import sos_planes
    q = 2 ** (bit - 1)
    split = np.float32(2.0 ** -split_exp)
    rng = np.random.default_rng(split_exp)
    A = torch.softmax(torch.from_numpy(rng.standard_normal((97, 197)).astype(np.float32) * 3), -1).numpy()
    below, above = np.nextafter(split, np.float32(0)), np.nextafter(split, np.float32(2))
    A[0, :8] = [split, below, above, 0.0, 1.0, split / 2, min(np.float32(1), split * 2), split * np.float32(0.999)]
    a_int = split / np.float32(q - 1)
    k = np.arange(30, dtype=np.float32)
    A[1, :30] = np.minimum((k + np.float32(0.5)) * a_int, np.float32(1))   # exact .5 cases of the low-range grid
    A[2, :30] = np.minimum((k + np.float32(0.5)) / np.float32(q - 1), np.float32(1))   # ... of the high-range grid
    ref_hi, ref_lo = sos_planes(A, split, q)
    hi, _ = eng.pack_plane_i8(_t(A), mode="sos_hi", scales=torch.tensor([split]), lo=0, hi=q - 1, qmax=q)
    lo, _ = eng.pack_plane_i8(_t(A), mode="sos_lo", scales=torch.tensor([split]), lo=0, hi=q - 1, qmax=q)
    np.testing.assert_array_equal(hi.cpu().numpy(), ref_hi)
    np.testing.assert_array_equal(lo.cpu().numpy(), ref_lo)
    # App. A-9, spelled out: above the split the low plane saturates, below it the high plane is a constant
    assert (ref_lo[A >= split] == ref_lo[0, 0]).all()
    assert (ref_hi[A < split] == int(np.rint(split * np.float32(q - 1)))).all()
    if float(split) * (q - 1) < 0.5:
        assert (ref_hi[A < split] == 0).all()


@pytest.mark.parametrize("lo,hi", [(-128, 127), (-32, 31), (0, 127), (-128, 0), (-8, 7)])
def test_symmetric_plane_and_fake_quant_bit_exact(eng, lo, hi):
    """clamp(rint(x/s), lo, hi) with per-row-block scales (the weight planes' layout) and p4v_fake_quant (= index * s)."""
    from oracle.ptq4vit_oracle import fake_quant, quant_int
    rng = np.random.default_rng(abs(lo) + hi)
    x = (rng.standard_normal((300, 130)) * 2).astype(np.float32)
    s = np.array([0.0371, 0.011, 0.5], dtype=np.float32)
    x[0, :100] = (np.arange(100, dtype=np.float32) - 50 + 0.5) * s[0]
    rows_s = np.repeat(s, 100)[:, None]
    plane, _ = eng.pack_plane_i8(_t(x), mode="sym", scales=torch.from_numpy(s), rows_per_scale=100, lo=lo, hi=hi,
                                 qmax=max(-lo, hi + 1))
    np.testing.assert_array_equal(plane.cpu().numpy(), quant_int(x, rows_s, lo, hi))
    y = eng.fake_quant(_t(x), torch.from_numpy(s), 100, lo, hi)
    np.testing.assert_array_equal(y.cpu().numpy(), fake_quant(x, rows_s, lo, hi))


def test_sat8_quantiser_is_the_ieee_division_on_and_around_every_breakpoint(eng):
    """quant16_sat8 (round 5: fma + v_cvt_pk_u8_f32 twice, 4.5 operations per element, csrc/p4v_kernels.h) against numpy's IEEE
    division on the inputs built to break it: for seven scales (a power of two, values whose reciprocal rounds up / down, tiny,
    huge) every breakpoint (k + 0.5) s of the 8-bit grid, its two fp32 neighbours on either side, the saturation ends and far
    beyond, zeros, and 200 000 random values (1.2e-4 of them fall into the quantiser's flagged band and take the division).
    The same plane with the quantiser switched off (tuning 12 = 11: quant_fast1) must be identical too."""
    from oracle.ptq4vit_oracle import quant_int
    rng = np.random.default_rng(5)
    scales = np.array([0.25, 0.0371, 1.0 / 3.0, 0.011, 3e-7, 1.7e5, 0.7], dtype=np.float32)
    rows = []
    for s in scales:
        k = np.arange(-135, 135, dtype=np.float32)
        b = ((k + np.float32(0.5)) * s).astype(np.float32)
        cases = [b]
        for step in (1, 2):
            lo_n, hi_n = b.copy(), b.copy()
            for _ in range(step):
                lo_n, hi_n = np.nextafter(lo_n, np.float32(-np.inf)), np.nextafter(hi_n, np.float32(np.inf))
            cases += [lo_n, hi_n]
        cases.append(np.array([0.0, -0.0, 127 * s, 128 * s, -128 * s, -129 * s, 1e30, -1e30, 3e38, -3e38, s, -s, s / 2, -s / 2], dtype=np.float32))
        cases.append((rng.standard_normal(200000) * 60 * s).astype(np.float32))
        row = np.concatenate(cases)
        rows.append(row)
    n = max(len(r) for r in rows)
    n = (n + 15) // 16 * 16
    x = np.zeros((len(scales), n), dtype=np.float32)
    for i, r in enumerate(rows):
        x[i, :len(r)] = r
    with np.errstate(over="ignore"):                      # 3e38 / 3e-7 overflows to inf on purpose: both sides must saturate
        want = quant_int(x, scales[:, None], -128, 127)
    got, _ = eng.pack_plane_i8(_t(x), mode="sym", scales=torch.from_numpy(scales), rows_per_scale=1, lo=-128, hi=127, qmax=128)
    np.testing.assert_array_equal(got.cpu().numpy()[:, :n], want)
    try:
        eng.debug_tuning(12, 11)
        old, _ = eng.pack_plane_i8(_t(x), mode="sym", scales=torch.from_numpy(scales), rows_per_scale=1, lo=-128, hi=127, qmax=128)
    finally:
        eng.debug_tuning(12, 0)
    np.testing.assert_array_equal(old.cpu().numpy()[:, :n], want)


# ---- integer export (row f-3) -----------------------------------------------------------------------------------------
def _calibrated_mini_on_gpu():
    from ptq4vit_amd.configs import PTQ4ViT
    from ptq4vit_amd.utils import models, net_wrap
    g = np.load("tests/golden/minivit_ptq4vit.npz", allow_pickle=False)
    kw = json.loads(str(g["model_kwargs"]))
    net = models.get_net("vit_tiny_patch16_224", seed=0, device="cuda", **kw)
    wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
    for n, m in wrapped.items():
        key = n.replace(".", "__")
        for a in ("w_interval", "a_interval", "A_interval", "B_interval", "split"):
            if f"{key}::{a}" in g.files:
                setattr(m, a, torch.from_numpy(g[f"{key}::{a}"]).cuda())
        m.calibrated = True
    return g, net, wrapped


def test_integer_export_on_gpu_matches_the_reference_fixture():
    """Every weight image and every activation image of tests/golden/minivit_integer.npz (made by the reference's
    utils/integer.py) reproduced on the GPU by p4v_export_quantize: int8 / twin-uint8, bit for bit."""
    from ptq4vit_amd.quant_layers.matmul import MinMaxQuantMatMul
    from ptq4vit_amd.utils import integer
    g, net, wrapped = _calibrated_mini_on_gpu()
    gi = np.load("tests/golden/minivit_integer.npz", allow_pickle=False)
    ws = integer.get_model_int_weight(wrapped)
    expect = {k[:-len("::w_int")].replace("__", ".") for k in gi.files if k.endswith("::w_int")}
    assert set(ws) == expect and len(ws) > 0
    for n, w_int in ws.items():
        key = n.replace(".", "__")
        assert w_int.dtype == torch.int8 and not w_int.is_cuda
        np.testing.assert_array_equal(w_int.numpy(), gi[f"{key}::w_int"], err_msg=n)
    checked = 0
    for n, m in wrapped.items():
        key = n.replace(".", "__")
        if f"{key}::int_input0" not in gi.files:
            continue
        if isinstance(m, MinMaxQuantMatMul):
            B = _t(g[f"{key}::B"])
            if "matmul1" in n:
                B = B.transpose(-2, -1).contiguous().transpose(-2, -1)     # a k.transpose view, as the model passes it
            inputs = (_t(g[f"{key}::A"]), B)
        else:
            inputs = (_t(g[f"{key}::x"]),)
        integer.quantize_int_activation(m, inputs)
        for i, t in enumerate(m.int_input):
            ref = gi[f"{key}::int_input{i}"]
            assert str(t.dtype).replace("torch.", "") == str(ref.dtype) and not t.is_cuda
            np.testing.assert_array_equal(t.numpy(), ref, err_msg=f"{n} input {i}")
            checked += 1
    assert checked >= 14
    # the public float-index helper on GPU operands, ragged blocks (padding view of integer.py:28-43)
    A = torch.randn(2, 5, 7, 9, device="cuda")
    iv = torch.rand(1, 2, 1, 3, 1, 2, 1, device="cuda") + 0.05
    got = integer.quantize_matmul_input(A, iv, 128, 2, 3, 2, 3, 3, 5)
    want = integer.quantize_matmul_input(A.cpu(), iv.cpu(), 128, 2, 3, 2, 3, 3, 5)
    xp = torch.nn.functional.pad(A.cpu(), [0, 1, 0, 2, 0, 1]).view(-1, 2, 3, 3, 3, 2, 5)
    ref = (xp / iv.cpu()).round().clamp(-128, 127).view(-1, 6, 9, 10)[:, :5, :7, :9]
    assert got.is_cuda and torch.equal(got.cpu(), want) and torch.equal(want, ref)


# ---- quant_forward vs the reference's own outputs (row f-2) -------------------------------------------------------------
def _close_to_reference(got, ref, what, grid):
    """The reference multiplies fake-quantised fp32 operands with an fp32 GEMM; the engine multiplies the grid indices
    exactly (int32) and rescales once.  Same real-number value, different rounding: agreement to fp32 GEMM noise."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    scale = np.abs(ref).max()
    err = np.abs(got - ref).max() / scale
    print(f"[quant_forward] {what}: max |diff| / max |ref| = {err:.2e}")
    assert err <= 2e-6 * grid, f"{what}: {err:.3e}"


@pytest.mark.parametrize("name", golden_names("linear_") + golden_names("postgelu_"))
def test_linear_quant_forward_vs_reference_output(eng, name):
    g = load_golden(name)
    p = g["params"]
    out = eng.linear_quant_forward(weight=_t(g["weight"]), bias=_t(g["bias"]) if "bias" in g else None, x=_t(g["x"]),
                                   w_interval=_t(g["w_interval"]), a_interval=_t(g["a_interval"]), w_bit=p["w_bit"],
                                   a_bit=p["a_bit"], n_V=p["n_V"], n_H=p.get("n_H", 1), n_a=p.get("n_a", 1),
                                   postgelu=p["postgelu"])
    _close_to_reference(out.cpu().numpy(), g["quant_forward"], name, g["weight"].shape[1] ** 0.5)


@pytest.mark.parametrize("name", golden_names("matmul_"))
def test_matmul_quant_forward_vs_reference_output(eng, name):
    g = load_golden(name)
    p = g["params"]
    out = eng.matmul_quant_forward(A=_t(g["A"]), B=_t(g["B"]), A_interval=_t(np.asarray(g["A_interval"])),
                                   B_interval=_t(g["B_interval"]), split=_t(np.asarray(g["split"])) if p["sos"] else None,
                                   A_bit=p["A_bit"], B_bit=p["B_bit"], sos=p["sos"])
    _close_to_reference(out.cpu().numpy(), g["quant_forward"], name, g["A"].shape[-1] ** 0.5)


def test_minivit_quantised_logits_vs_reference():
    """Stand-in for the reference's post-quant accuracy check (example/test_vit.py:26-45; no ImageNet here): the mini ViT
    calibrated BY THE REFERENCE (intervals from the fixture), run in quant_forward mode on the GPU -- Linear and both
    MatMuls on the int8 MFMA path -- reproduces the reference's quantised logits."""
    from ptq4vit_amd.quant_layers.conv import MinMaxQuantConv2d
    g, net, wrapped = _calibrated_mini_on_gpu()
    for m in wrapped.values():
        m.mode = "quant_forward"
    images = _t(g["images"])
    with torch.no_grad():
        logits = net(images).cpu().numpy()
    ref = g["quant_logits"]
    err = np.abs(logits - ref).max() / np.abs(ref).max()
    print(f"[quant_logits] int8 path vs reference: max |diff| / max |ref| = {err:.2e}; "
          f"argmax agreement {int((logits.argmax(1) == ref.argmax(1)).sum())}/{len(ref)}")
    # a grid index that flips in one layer (operands differing in the last fp32 bit) moves that activation by one
    # interval; two blocks deep this stays far below the quantisation error itself (raw vs quantised logits: ~4e-2)
    assert err <= 2e-3
    assert (logits.argmax(1) == ref.argmax(1)).all()
    # and the fake-quant fp32 formulation of the same modules (the reference's arithmetic) agrees more tightly still
    for m in wrapped.values():
        if hasattr(m, "int8_forward"):
            m.int8_forward = False
    with torch.no_grad():
        logits_fq = net(images).cpu().numpy()
    err_fq = np.abs(logits_fq - ref).max() / np.abs(ref).max()
    print(f"[quant_logits] fp32 fake-quant path vs reference: {err_fq:.2e}")
    assert err_fq <= 2e-3
    assert any(isinstance(m, MinMaxQuantConv2d) for m in wrapped.values())


# ---- the capture pass's append: many (tensor -> slot of its cache) copies in one launch -----------------------------------
@pytest.mark.parametrize("index", [0, 2])
def test_multi_copy_appends_every_block_in_one_launch(eng, index):
    """p4v_multi_copy: block t of the table goes to dst base + index * bytes, bit-exact, nothing else written; sizes that
    are whole 16-byte runs, ragged (4-byte path), one element, and a source whose address is only 4-byte aligned."""
    g = torch.Generator().manual_seed(11)
    sizes = [4096, 1000003, 1, 7, 16 * 333, 250001]
    backing = torch.randn(sum(sizes) + len(sizes) + 1, generator=g).cuda()
    srcs, off = [], 1                                     # start 4 bytes into the allocation: not 16-byte aligned
    for s in sizes:
        srcs.append(backing[off:off + s])
        off += s + 1
    dsts = [torch.full((3 * s,), -7.0, device="cuda") for s in sizes]
    rows = [[s.data_ptr(), d.data_ptr(), s.numel() * 4] for s, d in zip(srcs, dsts)]
    table = torch.tensor(rows, dtype=torch.int64).cuda()
    eng.multi_copy(table, len(rows), index, max(r[2] for r in rows), torch.device("cuda:0"))
    torch.cuda.synchronize()
    for s, d in zip(srcs, dsts):
        n = s.numel()
        assert torch.equal(d[index * n:(index + 1) * n], s)
        rest = torch.cat([d[:index * n], d[(index + 1) * n:]])
        assert bool((rest == -7.0).all())


# ---- PTQSLQuantConv2d's own, non-batching search ---------------------------------------------------
@pytest.mark.parametrize("name", golden_names("ptqslconv_"))
def test_ptqsl_conv_own_search_vs_reference(name):
    """PTQSLQuantConv2d.calibration_step2(x) (reference conv.py:253-277) through the GPU Linear engine on the unfolded
    patches: intervals against what the REFERENCE selected on the same tensors (bit-identical, or another entry of the same
    candidate table at a near-tie), the returned quantised output against the reference's.  The reference's cosine runs over
    all output channels whatever the block, so with n_V / n_H > 1 it is refused (the oracle covers that fixture on the CPU)."""
    from tests.helpers import assert_on_candidate_grid, candidate_grid
    from ptq4vit_amd.quant_layers.conv import PTQSLQuantConv2d
    g = load_golden(name)
    p = dict(g["params"])
    p.pop("kind")
    st = p.pop("stride")
    oc, ic, k, _ = g["weight"].shape
    m = PTQSLQuantConv2d(ic, oc, k, st, parallel_eq_n=10, **p).cuda()
    m.weight.data = _t(g["weight"])
    m.bias.data = _t(g["bias"])
    x = _t(g["x"])
    m.raw_input, m.raw_out = x, _t(g["out"])
    m.raw_grad = _t(g["grad"]) if p["metric"] == "hessian" else None
    if p["metric"] == "cosine" and (p["n_V"] > 1 or p["n_H"] > 1):
        with pytest.raises(NotImplementedError):
            m.calibration_step2(x)
        return
    with torch.no_grad():
        out = m.calibration_step2(x)
    torch.cuda.synchronize()
    assert m.calibrated and not hasattr(m, "raw_out")
    assert tuple(m.w_interval.shape) == (p["n_V"], 1, p["n_H"], 1) and m.a_interval.dim() == 0
    mult = candidate_grid(p["eq_alpha"], p["eq_beta"], p["eq_n"])
    moved = assert_on_candidate_grid(m.w_interval.cpu().numpy(), g["w_interval"], mult, name + " w_interval")
    moved += assert_on_candidate_grid(m.a_interval.cpu().numpy(), g["a_interval"], mult, name + " a_interval")
    print(f"[parity] {name}: {moved} intervals on another entry of the candidate table (near-ties), the rest bit-identical")
    if moved == 0:
        np.testing.assert_allclose(out.cpu().numpy(), g["quant_forward"], rtol=1e-4, atol=1e-5)
        m.mode = "quant_forward"
        with torch.no_grad():
            np.testing.assert_allclose(m(x).cpu().numpy(), g["quant_forward"], rtol=1e-4, atol=1e-5)
