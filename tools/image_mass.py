"""Per-IMAGE share of the Hessian weight raw_grad^2 in a ViT-B/224 x 32 calibration: would a slice of the heaviest images do for
stage A of the attention matmuls?  python tools/image_mass.py"""
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
rows = {}
for n, m in wrapped.items():
    orig = m.calibration_step2
    def rec(_o=orig, _m=m, _n=n):
        g = _m.raw_grad.double() ** 2
        per_img = g.reshape(g.shape[0], -1).sum(1)
        s, _ = torch.sort(per_img, descending=True)
        cum = (torch.cumsum(s, 0) / s.sum()).cpu()
        extra = ""
        if g.dim() == 4 and g.shape[1] == 12:            # matmul: top-16 rows of the chosen images vs everything
            rm = g.sum(-1)                               # [b, H, M]
            top16 = torch.topk(rm, 16, dim=-1).values.sum(-1)          # [b, H]
            order = torch.argsort(per_img, descending=True)
            extra = "  16 rows of top-8 images: %.4f  of top-4: %.4f" % (float(top16[order[:8]].sum() / g.sum()), float(top16[order[:4]].sum() / g.sum()))
        rows[_n] = "top 1/2/4/8/16 images: " + " ".join("%.4f" % float(cum[k - 1]) for k in (1, 2, 4, 8, 16)) + extra
        return _o()
    m.calibration_step2 = rec
with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
    HessianQuantCalibrator(net, wrapped, L(), sequential=False, batch_size=4).batching_quant_calib()
for n in list(rows)[:1] + [k for k in rows if k.startswith("blocks.0.") or k.startswith("blocks.5.") or k.startswith("blocks.11.")] + ["head"]:
    print(f"{n:28s} {rows[n]}")
