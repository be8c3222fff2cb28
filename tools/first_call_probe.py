"""Where the first calibration of a process spends its time (bench.py reports it as first_calibration_s)."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
t0 = time.time()
from ptq4vit_amd import engine
from ptq4vit_amd.configs import PTQ4ViT
from ptq4vit_amd.utils import models, net_wrap
from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator
dev = torch.device("cuda:0")
net = models.get_net("vit_base_patch16_224", seed=0, device=dev)
with contextlib.redirect_stdout(io.StringIO()):
    wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
images = torch.randn(32, 3, 224, 224, generator=torch.Generator().manual_seed(0)).to(dev)
torch.cuda.synchronize(); print(f"setup {time.time() - t0:.2f} s")

class L:
    batch_size = 32
    def __iter__(self):
        yield images, None

for i in range(3):
    for m in wrapped.values():
        m.mode = "raw"
    cal = HessianQuantCalibrator(net, wrapped, L(), sequential=False, batch_size=4)
    t = time.time()
    sm = cal._raw_pred_softmax(); torch.cuda.synchronize(); t1 = time.time()
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        cal.batching_quant_calib()
    torch.cuda.synchronize(); t2 = time.time()
    print(f"calibration {i}: raw_pred {t1 - t:.3f} s (then again inside), total {t2 - t1:.3f} s: {cal.timings}")
