#!/usr/bin/env python
"""FP32 and post-quantisation ImageNet top-1 of one model -- the accuracy half of BASELINE.json's north_star ("top-1 within
+-0.1 % of the reference on identical calibration images"), as the reference's experiment does it:

    net = get_net(name)                                   reference example/test_vit.py:98  (timm weights, utils/models.py:77)
    wrapped = wrap_modules_in_net(net, cfg)               test_vit.py:100
    calib_loader = g.calib_loader(num=32)                 test_vit.py:102   (seed 3, utils/datasets.py:88-94)
    HessianQuantCalibrator(net, wrapped, calib_loader, sequential=False, batch_size=4).batching_quant_calib()   :104-105
    acc = test_classification(net, test_loader)           test_vit.py:107, 26-45

    python tools/eval_top1.py --imagenet /datasets/imagenet --weights vit_base_patch16_224.pth --model vit_base_patch16_224
        [--config PTQ4ViT|BasePTQ] [--bits 8] [--calib 32] [--calib-seed 3] [--batch 128] [--max-val N] [--workers 8]
        [--save-intervals FILE] [--json OUT]

Needs a GPU (the calibration engine has no CPU path), an ImageNet root with `train/` and `val/` class folders and a timm
checkpoint of the model (`timm.create_model(name, pretrained=True).state_dict()` saved with torch.save, or the .safetensors
file timm downloads).  Neither exists in the build environment, so the numbers this prints cannot be produced there; the tool
itself is exercised end to end on a synthetic ImageFolder by tests/test_top1_harness.py.
Prints ONE JSON line: {"model", "config", "bits", "calib_images", "fp32_top1", "quant_top1", "drop", "val_images",
"calibration_s", ...}.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def set_bits(cfg, bits):
    """W{bits}A{bits} on every wrapped module type (reference example/test_all.py:100-106 edits the config the same way)."""
    cfg.bit = bits
    for tab in (cfg.w_bit, cfg.a_bit, cfg.A_bit, cfg.B_bit):
        for k in tab:
            tab[k] = bits
    # the patch embedding keeps its fp32 input in both shipped configurations (configs/PTQ4ViT.py:54 sets a_bit 32 in get_module)


def evaluate(model, imagenet, weights=None, config="PTQ4ViT", bits=8, calib=32, calib_seed=3, batch=128, max_val=None,
             workers=0, save_intervals=None, device="cuda", quiet=False):
    """Returns the result dict (see the module docstring).  `weights=None` keeps the seeded random initialisation (tests)."""
    import contextlib
    import importlib
    import io

    import torch

    from ptq4vit_amd.utils import datasets, models, net_wrap
    from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator

    cfg = importlib.import_module(f"ptq4vit_amd.configs.{config}")
    saved = (cfg.bit, dict(cfg.w_bit), dict(cfg.a_bit), dict(cfg.A_bit), dict(cfg.B_bit))
    net = models.get_net(model, device=device)
    info = {"missing_buffers": [], "unexpected_keys": []}
    if weights:
        info["missing_buffers"], info["unexpected_keys"] = models.load_pretrained(net, weights)
    g = datasets.ViTImageNetLoaderGenerator(imagenet, "imagenet", batch, batch, workers, kwargs={"model": net})
    test_loader = g.test_loader()
    max_it = None if max_val is None else max(1, (max_val + batch - 1) // batch)
    t0 = time.time()
    fp32 = datasets.test_classification(net, test_loader, max_iteration=max_it, description=None if quiet else f"{model} fp32")
    t_fp = time.time() - t0
    try:
        set_bits(cfg, bits)
        with contextlib.redirect_stdout(io.StringIO()):
            wrapped = net_wrap.wrap_modules_in_net(net, cfg)
    finally:
        cfg.bit = saved[0]
        for tab, old in zip((cfg.w_bit, cfg.a_bit, cfg.A_bit, cfg.B_bit), saved[1:]):
            tab.clear()
            tab.update(old)
    calib_loader = g.calib_loader(num=calib, seed=calib_seed)
    t0 = time.time()
    cal = HessianQuantCalibrator(net, wrapped, calib_loader, sequential=False, batch_size=4)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        cal.batching_quant_calib()
    torch.cuda.synchronize()
    t_cal = time.time() - t0
    if save_intervals:
        from ptq4vit_amd.utils.intervals import save_intervals as _save
        _save(wrapped, save_intervals, meta={"model": model, "config": config, "bits": bits, "calib_images": calib,
                                              "calib_seed": calib_seed})
    t0 = time.time()
    quant = datasets.test_classification(net, test_loader, max_iteration=max_it, description=None if quiet else f"{model} W{bits}A{bits}")
    t_q = time.time() - t0
    n_val = min(len(g.test_set), (max_it or 10 ** 12) * batch)
    return {"model": model, "config": config, "bits": bits, "calib_images": calib, "calib_seed": calib_seed,
            "calib_indices_head": [int(i) for i in g.calib_indices(calib, calib_seed)[:8]],
            "fp32_top1": fp32, "quant_top1": quant, "drop": fp32 - quant, "val_images": n_val, "wrapped_modules": len(wrapped),
            "calibration_s": t_cal, "capture_s": cal.timings.get("capture_s"), "search_s": cal.timings.get("search_s"),
            "fp32_eval_s": t_fp, "quant_eval_s": t_q, "weights": weights or "random init (seed 0)", **info}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--imagenet", required=True, help="ImageNet root with train/ and val/ class folders")
    ap.add_argument("--weights", default=None, help="timm checkpoint (.pth state dict or .safetensors); default: random init")
    ap.add_argument("--model", default="vit_base_patch16_224")
    ap.add_argument("--config", default="PTQ4ViT", choices=["PTQ4ViT", "BasePTQ"])
    ap.add_argument("--bits", type=int, default=8)
    ap.add_argument("--calib", type=int, default=32)
    ap.add_argument("--calib-seed", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--max-val", type=int, default=None, help="evaluate only the first N validation images")
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--save-intervals", default=None, help="write the calibrated intervals (utils/intervals.py) to this file")
    ap.add_argument("--json", default=None, help="also write the result line to this file")
    a = ap.parse_args(argv)
    import ptq4vit_amd
    ptq4vit_amd.configure_runtime()          # GPU_MAX_HW_QUEUES for the search streams, before the first GPU call
    res = evaluate(a.model, a.imagenet, a.weights, a.config, a.bits, a.calib, a.calib_seed, a.batch, a.max_val, a.workers,
                   a.save_intervals)
    line = json.dumps(res)
    print(line)
    if a.json:
        with open(a.json, "w") as fh:
            fh.write(line + "\n")
    return res


if __name__ == "__main__":
    main()
