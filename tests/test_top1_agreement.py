"""Top-1 bookkeeping pinned on what exists offline (north_star: post-quant top-1 within +-0.1 % of the reference).

ImageNet and pretrained weights are not available in the build or on the GPU box, so the number itself cannot be measured.
What can: the reference's evaluation loop (example/test_vit.py:26-45 -- argmax of the QUANTISED network's logits per image)
was run BY THE REFERENCE (oracle/gen_golden.py::gen_deit_tiny_eval, imported from /root/reference in the build container) on
DeiT-tiny/224 BasePTQ, calibrated by the reference on 4 seeded images, over 1 000 seeded evaluation images.  The fixture
(tests/golden/deit_tiny_224_baseptq_eval1000.npz) holds the reference's top-1 prediction of every image, its top-1 / top-2
margin, and the raw network's prediction (the label a data-free top-1 uses: the reference's quantised network agrees with the raw
one on 933 / 1 000 -- random weights, noise images: the margins are of the size of the quantisation error).

  * CPU: this repository's module classes with the REFERENCE's intervals installed reproduce the reference's prediction on every
    one of the 1 000 images (fp32 fake-quant forward, the arithmetic of linear.py:62-67 / matmul.py:140-145 / conv.py:609-614).
  * GPU: every module of the network through the int8 MFMA `quant_forward` against the CPU module on the same input (fp32
    rounding); then top-1 against the raw network's labels for the reference-calibrated and the ENGINE-calibrated network next
    to the reference's 93.3 % (image-by-image equality across machines is not a property a re-quantising network has: see the test).

The day P4V_IMAGENET / P4V_WEIGHTS exist (tools/eval_top1.py) the only unknown left is the data.
"""
import contextlib
import io

import numpy as np
import pytest
import torch

EVAL = "tests/golden/deit_tiny_224_baseptq_eval1000.npz"
CALIB = "tests/golden/deit_tiny_224_baseptq_4img.npz"


def _net_with_reference_intervals(device):
    from ptq4vit_amd.configs import BasePTQ
    from ptq4vit_amd.utils import models, net_wrap
    from ptq4vit_amd.utils.intervals import install_intervals
    g = np.load(CALIB, allow_pickle=False)
    net = models.get_net("deit_tiny_patch16_224", seed=0, device=device)
    with contextlib.redirect_stdout(io.StringIO()):
        wrapped = net_wrap.wrap_modules_in_net(net, BasePTQ)
    assert list(wrapped) == [str(n) for n in g["names"]]
    for n, m in wrapped.items():
        key = n.replace(".", "__")
        vals = {a: torch.from_numpy(g[f"{key}::{a}"]) for a in ("w_interval", "a_interval", "A_interval", "B_interval") if f"{key}::{a}" in g.files}
        install_intervals(m, vals, device=torch.device(device))
        m.mode = "quant_forward"
    return net, wrapped


def _eval_images(fx):
    import json
    cfg = json.loads(str(fx["config"]))
    ev = torch.randn(cfg["eval_images"], 3, 224, 224, generator=torch.Generator().manual_seed(cfg["eval_seed"]))
    assert abs(ev.double().sum().item() - float(fx["eval_sum"])) <= 1e-9 * float(fx["eval_abs_sum"]), "not the images the reference evaluated"
    return ev


def _predict(net, ev, device, batch):
    out = []
    with torch.no_grad():
        for i in range(0, ev.shape[0], batch):
            out.append(net(ev[i:i + batch].to(device)).float().cpu())
    return torch.cat(out)


def test_reference_intervals_reproduce_the_reference_top1_on_1000_images_cpu():
    fx = np.load(EVAL, allow_pickle=False)
    net, _ = _net_with_reference_intervals("cpu")
    ev = _eval_images(fx)
    q = _predict(net, ev, "cpu", 50)
    rng = float(fx["logit_range"])
    head_err = float((q[:16] - torch.from_numpy(fx["quant_logits_head"])).abs().max()) / rng
    same = q.argmax(1).numpy() == fx["quant_argmax"]
    print(f"[top1] CPU, reference intervals: {int(same.sum())}/1000 predictions equal to the reference's, logits of the first 16 images "
          f"to {head_err:.1e} of the logit range")
    assert head_err <= 1e-5, head_err
    assert same.all(), f"{int((~same).sum())} predictions differ; smallest reference margin among them {fx['quant_margin'][~same].min():.2e}"


@pytest.mark.gpu
def test_int8_quant_forward_and_engine_calibration_against_the_reference_top1_on_1000_images():
    """(a) MODULE BY MODULE: each of the 74 wrapped modules of the GPU network (int8 MFMA quant_forward, reference intervals) fed
    the input the CPU network's module saw reproduces the CPU module's output to fp32 rounding (measured 3e-7 .. 8e-7 of the
    output range) -- the inference path is the reference's arithmetic at full DeiT-tiny size.
    (b) THE NETWORK cannot be compared image by image across machines: LayerNorm / softmax / GELU differ by ~1e-6 between the
    GPU and the CPU, the next module RE-QUANTISES its input, one activation that lands on the other side of a rounding boundary
    moves an output by a whole grid step (1.8e-3 of the output range after the first block's qkv, measured), and from the third
    block on the two runs differ by the quantisation error itself (2e-2 of the logit range; tools/diff_quant_forward.py) --
    as would the reference on a GPU against the reference on a CPU.  What is comparable is the statistic the reference reports
    (example/test_vit.py:26-45): top-1 over the evaluation set -- here against the raw network's predictions as labels -- for
    the reference-calibrated network on this GPU, the ENGINE-calibrated network on this GPU, and the reference's own run (93.3 %)."""
    from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator
    fx = np.load(EVAL, allow_pickle=False)
    ev = _eval_images(fx)
    ref_pred, margin, raw_label = fx["quant_argmax"].astype(np.int64), fx["quant_margin"], fx["raw_argmax"].astype(np.int64)
    rng = float(fx["logit_range"])
    # (a) every module alone, on the CPU network's own inputs
    net_c, wr_c = _net_with_reference_intervals("cpu")
    net, wrapped = _net_with_reference_intervals("cuda")
    seen = {}
    hooks = [m.register_forward_hook(lambda mod, inp, out, _n=n: seen.__setitem__(_n, ([t.detach().clone() for t in inp], out.detach().clone())))
             for n, m in wr_c.items()]
    with torch.no_grad():
        net_c(ev[:32])                      # (32 images: 6 304 token rows per Linear, 96 (image, head) matrices per attention matmul)
    for h in hooks:
        h.remove()
    worst = (0.0, "")
    for n, m in wrapped.items():
        ins, out_c = seen[n]
        with torch.no_grad():
            out_g = m(*[t.cuda() for t in ins]).float().cpu()
        err = float((out_g - out_c).abs().max()) / (float(out_c.abs().max()) + 1e-30)
        worst = max(worst, (err, n))
        assert err <= 1e-5, f"{n} ({type(m).__name__}): int8 quant_forward leaves the fp32 fake-quant forward by {err:.2e} of the output range"
    del net_c, wr_c, seen
    # (b) the network on this GPU: reference-calibrated, then engine-calibrated
    pred = _predict(net, ev, "cuda", 100).argmax(1).numpy()
    for m in wrapped.values():
        m.mode = "raw"
    raw_gpu = _predict(net, ev, "cuda", 100).argmax(1).numpy()
    images = torch.randn(4, 3, 224, 224, generator=torch.Generator().manual_seed(0)).cuda()

    class Loader:
        batch_size = 4

        def __iter__(self):
            yield images, torch.zeros(4, dtype=torch.long)

    with contextlib.redirect_stdout(io.StringIO()):
        HessianQuantCalibrator(net, wrapped, Loader(), sequential=False, batch_size=4).batching_quant_calib()
    mine = _predict(net, ev, "cuda", 100).argmax(1).numpy()
    acc_ref = float((ref_pred == raw_label).mean())            # the reference's own run (CPU)
    acc_gpu = float((pred == raw_gpu).mean())                  # the reference's intervals on this GPU
    acc_eng = float((mine == raw_gpu).mean())                  # the engine's intervals on this GPU
    wide = margin > 0.05 * rng
    print(f"[top1] DeiT-tiny/224 BasePTQ x4, 1000 images, labels = the raw network's predictions: reference run {100 * acc_ref:.1f} %, "
          f"reference intervals on this GPU {100 * acc_gpu:.1f} %, engine-calibrated {100 * acc_eng:.1f} %; same prediction as the reference's "
          f"run: {100 * float((pred == ref_pred).mean()):.1f} % / {100 * float((mine == ref_pred).mean()):.1f} % of the images "
          f"({100 * float((pred == ref_pred)[wide].mean()):.1f} % / {100 * float((mine == ref_pred)[wide].mean()):.1f} % of the {int(wide.sum())} with a margin above "
          f"5 % of the logit range); raw predictions GPU vs CPU {int((raw_gpu == raw_label).sum())}/1000; modules alone: worst {worst[0]:.1e} ({worst[1]})")
    # thresholds = what this measures on MI355X (reference run 93.3 %, this GPU 92.4 % / 92.5 %; same prediction as the reference's
    # run on 92.4 % / 93.7 % of the images, on 100 % of the wide-margin ones) minus a small slack: a regression of a dozen
    # images in the int8 forward or in the engine's calibration fails here
    assert (raw_gpu == raw_label).mean() >= 0.995
    assert abs(acc_gpu - acc_ref) <= 0.012 and abs(acc_eng - acc_ref) <= 0.012, (acc_ref, acc_gpu, acc_eng)
    assert (pred == ref_pred).mean() >= 0.915 and (mine == ref_pred).mean() >= 0.925
    assert (pred == ref_pred)[wide].mean() >= 0.99 and (mine == ref_pred)[wide].mean() >= 0.99


@pytest.mark.gpu
def test_the_int8_path_against_the_fp32_fake_quant_path_on_the_same_gpu_isolates_the_engine():
    """What the cross-machine comparison above cannot separate: the ENGINE's contribution to a prediction.  Same GPU, same
    intervals (the reference's), same LayerNorm / softmax / GELU kernels -- the only difference between the two networks is the
    arithmetic of the wrapped modules: `quant_forward` on the int8 MFMA path (integer accumulation, one fp32 scale) against the
    reference's fp32 fake-quant formulation F.linear(quant_input(x), *quant_weight_bias()) (reference linear.py:62-67,
    matmul.py:140-145), which the modules run when `int8_forward` is off.

    (a) IN THE NETWORK, MODULE BY MODULE (the isolation proper): every wrapped module of the int8 network, on the input it
        really received for 64 evaluation images, evaluated both ways: the two arithmetics agree to fp32 rounding of the fp32
        path's own GEMM (<= 1e-5 of the module's output range; the integer product is exact).
    (b) THE NETWORKS END TO END: that rounding noise (1e-6) moves an activation across a rounding boundary of the NEXT module
        now and then -- one grid step -- and the 12 blocks amplify it (measured: logits apart by 2e-3 median / 2e-2 max of the logit
        range, the quantisation error itself being 3.5e-2).  On this data-free set (random weights, noise images) the top-1 /
        top-2 margins are of that size, so ~5 % of the predictions flip (measured 951 / 1000 equal) -- every one of them a
        near-tie of the fp32 path itself (margin below the logit difference), none among the images with a clear margin.  The
        same happens between the reference on a GPU and the reference on a CPU; it bounds how far ANY two correct
        implementations agree image by image on this set."""
    fx = np.load(EVAL, allow_pickle=False)
    ev = _eval_images(fx)
    rng = float(fx["logit_range"])
    net, wrapped = _net_with_reference_intervals("cuda")
    for m in wrapped.values():
        assert getattr(m, "int8_forward", True)
    # (a) module by module inside the int8 network
    seen = {}
    hooks = [m.register_forward_hook(lambda mod, inp, out, _n=n: seen.__setitem__(_n, ([t.detach() for t in inp], out.detach())))
             for n, m in wrapped.items()]
    with torch.no_grad():
        net(ev[:64].cuda())
    for h in hooks:
        h.remove()
    worst_mod = (0.0, "")
    for n, m in wrapped.items():
        ins, out_i = seen[n]
        m.int8_forward = False                      # instance attribute: the class default stays
        with torch.no_grad():
            out_f = m(*ins)
        m.int8_forward = True
        err = float((out_i - out_f).abs().max()) / (float(out_f.abs().max()) + 1e-30)
        worst_mod = max(worst_mod, (err, n))
        assert err <= 1e-5, f"{n} ({type(m).__name__}): int8 quant_forward vs fp32 fake-quant forward on the network's own input: {err:.2e}"
    del seen
    # (b) end to end
    q_int = _predict(net, ev, "cuda", 100)
    for m in wrapped.values():
        m.int8_forward = False
    q_f32 = _predict(net, ev, "cuda", 100)
    for m in wrapped.values():
        m.mode = "raw"
    raw = _predict(net, ev, "cuda", 100)
    same = (q_int.argmax(1) == q_f32.argmax(1)).numpy()
    d = (q_int - q_f32).abs()
    noise = float((q_f32 - raw).abs().max()) / rng                 # the quantisation error itself, in logit ranges
    worst, med = float(d.max()) / rng, float(d.median()) / rng
    top2 = q_f32.topk(2, dim=1).values
    margin = ((top2[:, 0] - top2[:, 1]) / rng).numpy()
    flipped_margin = float(margin[~same].max()) if (~same).any() else 0.0
    clear = margin > 2.0 * worst
    print(f"[top1] int8 quant_forward vs fp32 fake-quant forward, same GPU, same (reference) intervals: modules inside the network (64 images) "
          f"agree to {worst_mod[0]:.1e} of their output range (worst: {worst_mod[1]}); end to end over 1000 images: same prediction on "
          f"{int(same.sum())}/1000, on {int(same[clear].sum())}/{int(clear.sum())} of the images whose fp32-path margin exceeds twice the largest logit "
          f"difference (largest margin among the flipped: {flipped_margin:.1e} of the logit range); logits differ by {med:.1e} (median) / "
          f"{worst:.1e} (max) of the logit range -- the quantisation error itself is {noise:.1e}")
    assert same.mean() >= 0.93, int(same.sum())
    assert same[clear].all()
    assert flipped_margin <= 2.0 * worst                            # a flipped prediction is a near-tie of the fp32 path itself
    assert worst <= 0.8 * noise and med <= 0.12 * noise, (worst, med, noise)
