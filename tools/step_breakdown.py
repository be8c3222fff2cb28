"""GPU: where the host time of ONE fresh-network calibration goes besides capture and search (the step of bench.py): wall clock of
the calibrator's phases, each bracketed by a device synchronisation -- so the sum is LARGER than an unbracketed step; what matters is
what shows up outside `_capture` and the search."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptq4vit_amd
ptq4vit_amd.configure_runtime()
import torch
from ptq4vit_amd.configs import PTQ4ViT
from ptq4vit_amd.utils import models, net_wrap
from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator

model = sys.argv[1] if len(sys.argv) > 1 else "vit_base_patch16_224"
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 32
images = torch.randn(n_img, 3, 224, 224, generator=torch.Generator().manual_seed(0)).cuda()


class Loader:
    batch_size = n_img

    def __iter__(self):
        yield images, None


acc = {}


def timed(cls, name):
    fn = getattr(cls, name)

    def wrapper(self, *a, **k):
        torch.cuda.synchronize()
        t = time.perf_counter()
        try:
            return fn(self, *a, **k)
        finally:
            torch.cuda.synchronize()
            acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t) * 1e3
    setattr(cls, name, wrapper)


for nm in ("_raw_pred_softmax", "_estimate_cache_bytes", "_capture", "_search_grouped", "_resolve_budget"):
    timed(HessianQuantCalibrator, nm)


def fresh():
    net = models.get_net(model, seed=0, device="cuda")
    with contextlib.redirect_stdout(io.StringIO()):
        wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
    torch.cuda.synchronize()
    acc.clear()
    t = time.perf_counter()
    cal = HessianQuantCalibrator(net, wrapped, Loader(), sequential=False, batch_size=4)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        cal.batching_quant_calib()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3, dict(acc), dict(cal.timings)


for _ in range(3):
    fresh()
for _ in range(3):
    wall, a, tm = fresh()
    rest = wall - sum(a.values())
    print(f"{model} x {n_img}: wall {wall:.1f} ms = " + " + ".join(f"{k} {v:.1f}" for k, v in a.items()) + f" + rest {rest:.1f}   (calibrator: capture {tm['capture_s'] * 1e3:.1f} search {tm['search_s'] * 1e3:.1f})")
