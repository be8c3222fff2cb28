"""Which candidate ranges survive the exact pruning in a real ViT-B/224 x 32 calibration (stderr lines of the engine)."""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ptq4vit_amd import engine
from ptq4vit_amd.configs import PTQ4ViT
from ptq4vit_amd.utils import models, net_wrap
from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator
model = sys.argv[1] if len(sys.argv) > 1 else "vit_base_patch16_224"
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
net = models.get_net(model, seed=0, device=dev)
with contextlib.redirect_stdout(io.StringIO()):
    wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
s = models.input_size(model)
images = torch.randn(n_img, 3, s, s, generator=torch.Generator().manual_seed(0)).to(dev)

class L:
    batch_size = n_img
    def __iter__(self):
        yield images, None

cal = HessianQuantCalibrator(net, wrapped, L(), sequential=False, batch_size=4)
cal.search_streams = 1
engine.debug_tuning(4, 1)
engine.stats_enable(True)
with contextlib.redirect_stdout(io.StringIO()):
    cal.batching_quant_calib()
engine.stats_enable(False)
