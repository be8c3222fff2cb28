"""GPU: successive FRESH networks of a 128-image configuration (Swin-B/384, ViT-B/384): wall clock of each calibration against the
calibrator's own capture / search split, allocator state before and after, and what the calibrator prints (group plans, out-of-memory
retries) -- where the time of a fresh-network step goes that `breakdown` does not show."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptq4vit_amd
ptq4vit_amd.configure_runtime()
import torch
from ptq4vit_amd.configs import PTQ4ViT
from ptq4vit_amd.utils import models, net_wrap
from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator

model = sys.argv[1] if len(sys.argv) > 1 else "swin_base_patch4_window12_384"
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 128
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 8
res = 384 if "384" in model else 224
images = torch.randn(n_img, 3, res, res, generator=torch.Generator().manual_seed(0)).cuda()
for name in list(PTQ4ViT.w_bit):
    PTQ4ViT.w_bit[name] = bits
for name in list(PTQ4ViT.a_bit):
    PTQ4ViT.a_bit[name] = bits
for name in list(PTQ4ViT.A_bit):
    PTQ4ViT.A_bit[name] = bits
for name in list(PTQ4ViT.B_bit):
    PTQ4ViT.B_bit[name] = bits


class Loader:
    batch_size = n_img

    def __iter__(self):
        yield images, None


def gib(x):
    return round(x / 2**30, 1)


for i in range(4):
    net = models.get_net(model, seed=0, device="cuda")
    wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    print(f"--- network {i}: free {gib(free0)} GiB, torch reserved {gib(torch.cuda.memory_reserved())} allocated {gib(torch.cuda.memory_allocated())}", flush=True)
    t = time.time()
    cal = HessianQuantCalibrator(net, wrapped, Loader(), sequential=False, batch_size=4)
    cal.batching_quant_calib()
    torch.cuda.synchronize()
    dt = time.time() - t
    print(f"    wall {dt:.2f} s; timings {dict((k, round(v, 3)) for k, v in cal.timings.items() if isinstance(v, float))}", flush=True)
    del net, wrapped, cal
