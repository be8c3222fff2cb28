#!/usr/bin/env python
"""Throughput of the calibrated network's quant_forward (the top-1 evaluation loop of the
reference, example/test_vit.py:26-45) next to the raw fp32 forward.

    python tools/bench_forward.py --model vit_base_patch16_224 --batch 128
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ptq4vit_amd.configs import PTQ4ViT  # noqa: E402
from ptq4vit_amd.utils import models, net_wrap  # noqa: E402
from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator  # noqa: E402


class Loader:
    def __init__(self, x):
        self.x = x

    def __iter__(self):
        yield self.x, torch.zeros(self.x.shape[0], dtype=torch.long, device=self.x.device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="vit_base_patch16_224")
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    net = models.get_net(a.model, seed=0, device=dev)
    wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
    img = models.input_size(a.model)
    g = torch.Generator(device="cpu").manual_seed(0)
    calib = torch.randn(32, 3, img, img, generator=g).to(dev)
    HessianQuantCalibrator(net, wrapped, Loader(calib), sequential=False, batch_size=4).batching_quant_calib()
    x = torch.randn(a.batch, 3, img, img, generator=g).to(dev)

    def timed(mode):
        for m in wrapped.values():
            m.mode = mode
        with torch.no_grad():
            y = net(x)
            torch.cuda.synchronize()
            t = time.time()
            for _ in range(a.reps):
                y = net(x)
            torch.cuda.synchronize()
        return (time.time() - t) / a.reps, y

    t_raw, y_raw = timed("raw")
    t_q, y_q = timed("quant_forward")
    print(f"{a.model} batch {a.batch}: raw fp32 {a.batch / t_raw:.0f} img/s ({t_raw * 1e3:.1f} ms), "
          f"quant_forward {a.batch / t_q:.0f} img/s ({t_q * 1e3:.1f} ms); "
          f"logit rel diff {((y_q - y_raw).norm() / y_raw.norm()).item():.3e}")


if __name__ == "__main__":
    main()
