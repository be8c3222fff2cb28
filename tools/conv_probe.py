import contextlib, io, os, sys
sys.path.insert(0, os.getcwd())
import ptq4vit_amd; ptq4vit_amd.configure_runtime()
import torch
from ptq4vit_amd import engine
from ptq4vit_amd.configs import PTQ4ViT
from ptq4vit_amd.utils import models, net_wrap
from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator
net = models.get_net("vit_base_patch16_224", seed=0, device="cuda")
with contextlib.redirect_stdout(io.StringIO()):
    wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
images = torch.randn(32, 3, 224, 224, generator=torch.Generator().manual_seed(0)).cuda()
class L:
    batch_size = 32
    def __iter__(self): yield images, None
only = {k: v for k, v in wrapped.items() if "patch_embed" in k}
cal = HessianQuantCalibrator(net, wrapped, L(), sequential=False, batch_size=4)
cal.search_streams = 1
# capture everything, then search only the conv with printing on
import types
orig = cal._search_concurrent
engine.stats_enable(True)      # stats mode prints the executed range per stage
engine.debug_tuning(4, 1)
with contextlib.redirect_stdout(io.StringIO()):
    cal.batching_quant_calib()
torch.cuda.synchronize()
