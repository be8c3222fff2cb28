"""GPU: what a FRESH network's first calibration costs with the eager capture (default below 24 sub-batches) against a capture
graph recorded on the spot (use_graph = True) with 1 / 2 / 3 lanes -- the region the reference times (example/test_all.py:31-34)."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptq4vit_amd
ptq4vit_amd.configure_runtime()
import torch
from ptq4vit_amd.configs import PTQ4ViT
from ptq4vit_amd.utils import models, net_wrap
from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator

images = torch.randn(32, 3, 224, 224, generator=torch.Generator().manual_seed(0)).cuda()


class Loader:
    batch_size = 32

    def __iter__(self):
        yield images, None


def fresh(use_graph, lanes):
    net = models.get_net("vit_base_patch16_224", seed=0, device="cuda")
    with contextlib.redirect_stdout(io.StringIO()):
        wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
    torch.cuda.synchronize()
    t = time.time()
    cal = HessianQuantCalibrator(net, wrapped, Loader(), sequential=False, batch_size=4)
    if use_graph is not None:
        cal.use_graph = use_graph
    if lanes:
        cal.capture_lanes = lanes
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        cal.batching_quant_calib()
    torch.cuda.synchronize()
    dt = time.time() - t
    tm = cal.timings
    del net, wrapped, cal
    torch.cuda.empty_cache()
    return dt, tm["capture_s"], tm["search_s"]


fresh(None, 0); fresh(True, 3)          # warm the process (libraries, kernels, allocator)
for label, ug, ln in [("eager (default)", None, 0), ("graph, 1 lane", True, 1), ("graph, 2 lanes", True, 2), ("graph, 3 lanes", True, 3),
                      ("eager (default)", None, 0), ("graph, 1 lane", True, 1)]:
    r = [fresh(ug, ln) for _ in range(3)]
    best = min(r)
    print(f"{label:18s} fresh-network calibration {best[0] * 1e3:7.1f} ms (capture {best[1] * 1e3:6.1f} + search {best[2] * 1e3:6.1f}); all: {[round(x[0] * 1e3, 1) for x in r]}")
