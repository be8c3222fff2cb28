"""Where the HOST time of one ViT-B/224 x 32 calibration goes (cProfile, one search stream): the C calls of the engine against
the Python around them.  python tools/host_profile.py"""
import contextlib, cProfile, io, os, pstats, sys, time
os.environ.setdefault("P4V_SEARCH_STREAMS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ptq4vit_amd.configs import PTQ4ViT
from ptq4vit_amd.utils import models, net_wrap
from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator

dev = torch.device("cuda:0")
net = models.get_net("vit_base_patch16_224", seed=0, device=dev)
with contextlib.redirect_stdout(io.StringIO()):
    wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
images = torch.randn(32, 3, 224, 224, generator=torch.Generator().manual_seed(0)).to(dev)


class L:
    batch_size = 32

    def __iter__(self):
        yield images, None


def step():
    for m in wrapped.values():
        m.mode = "raw"
        m.calibrated = False
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        HessianQuantCalibrator(net, wrapped, L(), sequential=False, batch_size=4).batching_quant_calib()
    torch.cuda.synchronize()


for _ in range(3):
    step()
t0 = time.time()
step()
print(f"one step, unprofiled: {1e3 * (time.time() - t0):.1f} ms")
pr = cProfile.Profile()
pr.enable()
step()
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(28)
print("\n".join(l[:150] for l in out.getvalue().splitlines()[:50]))
