"""GPU: SURVEY.md s8 row f-4 pinned to the REFERENCE -- the non-batching classes and the other calibrator entry points.

Fixtures (oracle/gen_golden.py::gen_f4_layers / gen_f4_calibrators, made by running the reference's own classes):
  ptqsllinear_* / ptqslpostgelu_*   PTQSLQuantLinear / PostGeluPTQSLQuantLinear.calibration_step2(x)   linear.py:94-347
  ptqslmatmul_* / ptqslsos_*        PTQSLQuantMatMul / SoSPTQSLQuantMatMul.calibration_step2(A, B)     matmul.py:62-388
  quantileconv_*                    QuantileQuantConv2d.calibration_step2(x)                           conv.py:91-124
  minivit_calibrator_{sequential,parallel,hessian}   QuantCalibrator.quant_calib (both modes), HessianQuantCalibrator.quant_calib
                                                     on the 2-block mini ViT                           quant_calib.py:28-171,216-298
Bar = the batching classes' bar: score tables within SCORE_RTOL of the reference's (the non-batching classes average over the
batch where the batching classes sum: the engine's tables are rescaled by that constant), selections equal or near-ties by the
reference's own scores, intervals bit-identical -- or, where a tie resolved differently, another entry of the same candidate
table, counted and printed; the value calibration_step2 returns (the quantised forward) against the reference's.
"""
import contextlib
import io
import json

import numpy as np
import pytest
import torch

from tests.helpers import (CAPTURE_TOL, assert_argmax_tie_aware, assert_on_candidate_grid, assert_scores_close, candidate_grid,
                           golden_names, grid_steps_between, load_golden)

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _tables(pairs, name):
    flips = 0
    for i, (got, ref) in enumerate(pairs):
        ref2 = ref.reshape(ref.shape[0], -1)
        got = np.asarray(got)[: ref2.shape[0], : ref2.shape[1]]
        assert_scores_close(got, ref2, what=f"{name}[{i}]")
        flips += assert_argmax_tie_aware(np.argmax(got, axis=0), ref2, what=f"{name}[{i}]")
    return flips


def _forward_close(got, want, name, exact):
    got, want = got.detach().cpu().numpy(), np.asarray(want)
    assert got.shape == want.shape, name
    tol = 2e-5 if exact else 2e-2           # same intervals: summation order only; a grid step apart: quantisation-sized
    assert np.abs(got - want).max() <= tol * np.abs(want).max() + 1e-7, f"{name}: quantised forward differs from the reference's"


@pytest.mark.parametrize("name", golden_names("ptqsllinear_") + golden_names("ptqslpostgelu_"))
def test_ptqsl_linear_module_vs_reference(name):
    from ptq4vit_amd import engine
    from ptq4vit_amd.quant_layers.linear import PostGeluPTQSLQuantLinear, PTQSLQuantLinear
    g = load_golden(name)
    p = dict(g["params"])
    p.pop("kind")
    oc, postgelu = p.pop("oc"), p.pop("postgelu")
    x = _t(g["x"])
    grad = _t(g["grad"]) if p["metric"] == "hessian" else None
    # (1) every score table, through the engine call the module makes
    nH, nA, R = p.get("n_H", 1), p.get("n_a", 1), p["search_round"]
    _, _, scores, best = engine.linear_calibrate(weight=_t(g["weight"]), bias=_t(g["bias"]) if "bias" in g else None, x=x,
                                                 out=_t(g["out"]), grad=grad, postgelu=postgelu, want_scores=True,
                                                 n_H=nH, n_a=nA, **{k: v for k, v in p.items() if k not in ("n_H", "n_a")})
    torch.cuda.synchronize()
    scores = scores.cpu().numpy() / g["x"].shape[0]          # sum over the batch -> the non-batching classes' mean
    per_round = nH + nA
    pairs = []
    for r in range(R):
        pairs.append((scores[r, 0], g["scores"][r * per_round]))
        pairs.append((scores[r, 1][:, :1], g["scores"][r * per_round + nH]))
    flips = _tables(pairs, name)
    # (2) the module: calibration_step2(x) -> intervals + the returned quantised forward
    m = (PostGeluPTQSLQuantLinear if postgelu else PTQSLQuantLinear)(g["x"].shape[-1], oc, bias="bias" in g, **p).cuda()
    m.weight.data = _t(g["weight"])
    if "bias" in g:
        m.bias.data = _t(g["bias"])
    m.raw_input, m.raw_out, m.raw_grad = x, _t(g["out"]), grad
    with torch.no_grad():
        qf = m.calibration_step2(x)
    assert m.calibrated and not hasattr(m, "raw_out")
    a_iv = m.a_interval[0] if postgelu else m.a_interval
    if postgelu:
        assert isinstance(m.a_interval, list) and float(m.a_interval[1]) == 0.16997124254703522 / m.a_qmax
    assert tuple(m.w_interval.shape) == g["w_interval"].shape and tuple(a_iv.shape) == g["a_interval"].shape
    mult = candidate_grid(p["eq_alpha"], p["eq_beta"], p["eq_n"])
    moved = (assert_on_candidate_grid(m.w_interval.cpu().numpy(), g["w_interval"], mult, name + " w_interval")
             + assert_on_candidate_grid(a_iv.cpu().numpy(), g["a_interval"], mult, name + " a_interval"))
    print(f"[parity] {name}: {flips} near-tie flips in the tables, {moved} intervals on another grid entry")
    if flips == 0 and nH == 1 and nA == 1:
        assert moved == 0
    _forward_close(qf, g["quant_forward"], name, exact=(moved == 0))


@pytest.mark.parametrize("name", golden_names("ptqslmatmul_") + golden_names("ptqslsos_"))
def test_ptqsl_matmul_module_vs_reference(name):
    from ptq4vit_amd import engine
    from ptq4vit_amd.quant_layers.matmul import PTQSLQuantMatMul, SoSPTQSLQuantMatMul
    g = load_golden(name)
    p = dict(g["params"])
    p.pop("kind")
    sos = p.pop("sos")
    A, B = _t(g["A"]), _t(g["B"])
    b, H = g["A"].shape[:2]
    m = (SoSPTQSLQuantMatMul if sos else PTQSLQuantMatMul)(**p)
    m.raw_input, m.raw_out = [A, B], _t(g["out"])
    m.raw_grad = _t(g["grad"]) if p["metric"] == "hessian" else None
    # record the per-head tables of every pass the module runs (they are folded to groups below, as the module does)
    seen = []
    orig = engine.MatMulStepper
    fused = engine.matmul_calibrate

    class Spy(orig):
        def search_A(self, *a, **k):
            r = super().search_A(*a, **k)
            seen.append(("A", r[1].cpu().numpy()))
            return r

        def search_B(self, *a, **k):
            r = super().search_B(*a, **k)
            seen.append(("B", r[1].cpu().numpy()))
            return r

        def search_split(self, *a, **k):
            r = super().search_split(want_scores=True)
            seen.append(("split", r[2].cpu().numpy()))
            return r

    headwise = p.get("n_G_B", 1) == H and (sos or p.get("n_G_A", 1) == H)
    try:
        engine.MatMulStepper = Spy
        with torch.no_grad():
            qf = m.calibration_step2(A, B)
    finally:
        engine.MatMulStepper = orig
    assert m.calibrated
    nGA, nGB = (1 if sos else p.get("n_G_A", 1)), p.get("n_G_B", 1)
    assert tuple(m.B_interval.shape) == (1, nGB, 1, 1, 1, 1, 1) == g["B_interval"].shape
    flips = 0
    if not headwise:
        assert len(seen) == len(g["scores"])
        pairs = []
        for (kind, tab), ref in zip(seen, g["scores"]):
            if kind == "split":
                pairs.append((tab / b, ref.reshape(-1, 1)))           # sum over the batch of means -> one mean (matmul.py:335)
                continue
            nG = nGA if kind == "A" else nGB
            crb = -(-H // nG)
            grp = np.zeros((tab.shape[0], nG), dtype=np.float64)
            for h in range(H):
                grp[:, h // crb] += tab[:, h]
            pairs.append((grp / (b * crb), ref))                      # matmul.py:199: mean over the crb heads incl. padding
        flips = _tables(pairs, name)
    else:
        # head-wise configuration: the module takes the fused call; tables from the same call with want_scores
        A_iv, B_iv, split, scores, best = fused(A=A, B=B, out=_t(g["out"]), grad=m_grad(g, p), sos=sos, want_scores=True,
                                                **{k: v for k, v in p.items() if not k.startswith("n_G")})
        scores = scores.cpu().numpy() / b
        pairs = []
        for r in range(p["search_round"]):
            ta, tb = g["scores"][2 * r], g["scores"][2 * r + 1]
            pairs.append((scores[r, 0][:20, :1], ta.reshape(-1, 1)) if sos else (scores[r, 0], ta))
            pairs.append((scores[r, 1], tb))
        flips = _tables(pairs, name)
    mult = candidate_grid(p["eq_alpha"], p["eq_beta"], p["eq_n"])
    moved = assert_on_candidate_grid(m.B_interval.cpu().numpy(), g["B_interval"], mult, name + " B_interval")
    if sos:
        assert float(m.split) == float(g["split"]) and float(m.A_interval) == float(g["A_interval"])
    else:
        assert tuple(m.A_interval.shape) == g["A_interval"].shape
        moved += assert_on_candidate_grid(m.A_interval.cpu().numpy(), g["A_interval"], mult, name + " A_interval")
    print(f"[parity] {name}: {flips} near-tie flips in the tables, {moved} intervals on another grid entry")
    if flips == 0:
        assert moved == 0
    _forward_close(qf, g["quant_forward"], name, exact=(moved == 0))


def m_grad(g, p):
    return _t(g["grad"]) if p["metric"] == "hessian" else None


def test_quantile_conv_vs_reference():
    """QuantileQuantConv2d (conv.py:91-124): torch.quantile of |W| and |x|; on the GPU `tensor / python scalar` is a
    multiplication by the reciprocal, hence one ulp of slack on the intervals."""
    from ptq4vit_amd.quant_layers.conv import QuantileQuantConv2d
    g = load_golden("quantileconv_w8a8")
    p = dict(g["params"])
    p.pop("kind")
    st = p.pop("stride")
    oc, ic, k, _ = g["weight"].shape
    m = QuantileQuantConv2d(ic, oc, k, st, **p).cuda()
    m.weight.data, m.bias.data = _t(g["weight"]), _t(g["bias"])
    with torch.no_grad():
        qf = m.calibration_step2(_t(g["x"]))
    for a in ("w_interval", "a_interval"):
        got, want = float(getattr(m, a)), float(g[a])
        assert abs(got - want) <= 2.4e-7 * want, (a, got, want)
    _forward_close(qf, g["quant_forward"], "quantileconv", exact=False)


@pytest.mark.parametrize("run", ["sequential", "parallel", "hessian"])
def test_calibrator_entry_points_vs_reference(run):
    """QuantCalibrator(sequential=True / False).quant_calib() and HessianQuantCalibrator.quant_calib() on the mini ViT with
    every module a non-batching class, against the reference's own run of the same entry point: all 14 modules' intervals
    (bit-identical or another entry of the candidate table; counted) and the quantised logits."""
    from ptq4vit_amd.quant_layers.conv import PTQSLQuantConv2d
    from ptq4vit_amd.quant_layers.linear import PostGeluPTQSLQuantLinear, PTQSLQuantLinear
    from ptq4vit_amd.quant_layers.matmul import PTQSLQuantMatMul, SoSPTQSLQuantMatMul
    from ptq4vit_amd.utils import models, net_wrap
    from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator, QuantCalibrator
    g = np.load(f"tests/golden/minivit_calibrator_{run}.npz", allow_pickle=False)
    kw, hp = json.loads(str(g["model_kwargs"])), json.loads(str(g["hp"]))

    class cfg:  # noqa: N801  (the factory of oracle/gen_golden.py::gen_f4_calibrators over this package's classes)
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

    net = models.get_net("vit_tiny_patch16_224", seed=0, device="cuda", **kw)
    with contextlib.redirect_stdout(io.StringIO()):
        wrapped = net_wrap.wrap_modules_in_net(net, cfg)
    assert list(wrapped) == [str(n) for n in g["names"]]
    images = torch.from_numpy(g["images"]).cuda()

    class Loader:
        batch_size = images.shape[0]

        def __iter__(self):
            yield images, torch.zeros(images.shape[0], dtype=torch.long)

    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        if run == "hessian":
            HessianQuantCalibrator(net, wrapped, Loader(), sequential=False, batch_size=4).quant_calib()
        else:
            QuantCalibrator(net, wrapped, Loader(), sequential=(run == "sequential")).quant_calib()
    assert all(m.mode == "quant_forward" and m.calibrated for m in wrapped.values())
    mult = candidate_grid(hp["eq_alpha"], hp["eq_beta"], hp["eq_n"])
    total = moved = rounded = 0
    for n, m in wrapped.items():
        key = n.replace(".", "__")
        for a in ("w_interval", "a_interval", "A_interval", "B_interval", "split"):
            if f"{key}::{a}" not in g.files:
                continue
            want = g[f"{key}::{a}"]
            v = getattr(m, a)
            got = torch.as_tensor(v[0] if isinstance(v, (list, tuple)) else v).detach().cpu().numpy()
            assert got.shape == want.shape, (n, a, got.shape, want.shape)
            if a == "split" or (a == "A_interval" and m.__class__.__name__.startswith("SoS")):
                total += 1
                moved += int(float(got) != float(want))
                assert float(got) in [2.0 ** -i for i in range(20)] or a == "A_interval"
                continue
            total += want.size
            # (no step bound: the tensors these searches ran on were captured on THIS GPU, the reference's on its CPU -- raw_grad of a
            # non-sequential calibration is rounding noise, DESIGN.md s9, so the two searches optimise different objectives; what is
            # asserted is that every interval is an entry of the candidate table and, below, how many moved)
            assert_on_candidate_grid(got, want, mult, f"{n}.{a}", tol=CAPTURE_TOL, max_steps=None)
            for x, y in zip(got.reshape(-1), want.reshape(-1)):
                if x != y:
                    # 0 steps: the SAME candidate of a table whose initial interval differs in the last bits
                    if grid_steps_between(x, y, mult, tol=CAPTURE_TOL) == 0:
                        rounded += 1
                    else:
                        moved += 1
        if f"{key}::a_neg_interval" in g.files:
            assert float(m.a_interval[1]) == float(g[f"{key}::a_neg_interval"])
    with torch.no_grad():
        q = net(images).cpu().numpy()
    rng = float(g["quant_logits"].max() - g["quant_logits"].min())
    err = np.abs(q - g["quant_logits"]).max() / rng
    print(f"[parity] {run}_quant_calib on the mini ViT vs the reference's run: {total - moved - rounded}/{total} intervals "
          f"bit-identical, {rounded} the same candidate within 4e-7 (input rounding), {moved} on another entry of the candidate "
          f"table; quantised logits {err:.2e} of the logit range")
    # (`rounded`: the captured tensors come from this GPU's fp32 GEMMs, the reference's from the CPU's -- a min-max that sits
    # on an element whose last bits differ moves the whole candidate table by those bits; the selected INDEX is what is compared)
    if run == "hessian":
        # the Hessian weights of a non-sequential calibration are rounding noise of (pass of 4 images) - (pass of all images)
        # (DESIGN.md s9), realised differently by this GPU's GEMMs and by the reference's CPU run: the searches weigh the
        # output error with different noise and may settle on different candidates.  What is pinned for this entry point is
        # the structure (every interval an entry of the reference's candidate table, same shapes, the reference's split) and
        # that the quantised network stays close to the reference's; the selections themselves are pinned by the L2 runs
        # above and by tests/test_hip_model.py on the reference's own captured tensors.
        assert err <= 5e-2, err
    else:
        assert moved <= 0.1 * total, (moved, total)
        assert err <= (1e-5 if moved == 0 else 2e-2), err
