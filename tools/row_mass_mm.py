"""Hessian weight concentration inside the attention matmuls of a real ViT-B/224 x 32 capture: rows (queries) per (image, head)."""
import contextlib, io, os, sys
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

seen = {}
for n, m in wrapped.items():
    def rec(_m=m, _n=n):
        g = _m.raw_grad.float()
        if g.dim() == 4 and not hasattr(_m, "weight"):
            w = (g * g).sum(-1)                       # (b, H, M)
            tot = w.sum()
            res = []
            for k in (1, 4, 16, 32):
                res.append(float(torch.topk(w, k, dim=-1).values.sum() / tot))
            flat = w.reshape(-1)
            kk = flat.numel() // 16
            res.append(float(torch.topk(flat, kk).values.sum() / tot))
            res.append(float(w[..., 0].sum() / tot))
            seen[_n] = res
        _m.calibrated = True
        for a in ("raw_input", "raw_out", "raw_grad"):
            delattr(_m, a)
    m.calibration_step2 = rec
with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
    HessianQuantCalibrator(net, wrapped, L(), sequential=False, batch_size=4).batching_quant_calib()
print("module: mass in the top 1 / 4 / 16 / 32 rows of every (image, head); in the global top 1/16 of all rows; in row 0 (class token)")
for n, v in seen.items():
    print(f"{n:26s} " + "  ".join(f"{x:.3f}" for x in v))
