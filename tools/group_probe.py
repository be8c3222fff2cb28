"""GPU: the grouped search (p4v_calibrate_group) of a ViT-B/224 x 32 calibration: wall-clock per calibration, launches asked for /
issued / issue rounds, for one or several concurrent group calls (P4V_GROUP_CALLS) against the per-module search on four streams.
  python tools/group_probe.py [calls ...]       e.g.  1 2 3 0   (0 = per-module search)"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptq4vit_amd
ptq4vit_amd.configure_runtime()
import torch
from ptq4vit_amd import engine
from ptq4vit_amd.configs import PTQ4ViT
from ptq4vit_amd.utils import models, net_wrap
from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator

model = os.environ.get("P4V_MODEL", "vit_base_patch16_224")
calib = int(os.environ.get("P4V_CALIB", "32"))
images = torch.randn(calib, 3, 224, 224, generator=torch.Generator().manual_seed(0)).cuda()


class Loader:
    batch_size = calib

    def __iter__(self):
        yield images, None


net = models.get_net(model, seed=0, device="cuda")
with contextlib.redirect_stdout(io.StringIO()):
    wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)


def calibrate(calls):
    for m in wrapped.values():
        m.mode = "raw"
    engine.launch_counters(reset=True)
    torch.cuda.synchronize()
    t = time.time()
    cal = HessianQuantCalibrator(net, wrapped, Loader(), sequential=False, batch_size=4)
    cal.search_grouped, cal.group_calls = calls > 0, max(1, calls)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        cal.batching_quant_calib()
    torch.cuda.synchronize()
    return time.time() - t, cal.timings, engine.launch_counters(reset=True)


for calls in [int(a) for a in sys.argv[1:]] or [1, 2, 0]:
    calibrate(calls); calibrate(calls)
    r = [calibrate(calls) for _ in range(5)]
    best = min(r, key=lambda x: x[0])
    print(f"group calls {calls}: {best[0] * 1e3:7.1f} ms per calibration (capture {best[1]['capture_s'] * 1e3:5.1f} + search {best[1]['search_s'] * 1e3:5.1f}); "
          f"launches asked {best[2]['asked']} issued {best[2]['issued']} rounds {best[2]['rounds']}; all {[round(x[0] * 1e3, 1) for x in r]}", flush=True)
