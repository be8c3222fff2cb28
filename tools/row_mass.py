"""How concentrated is the Hessian weight (raw_grad^2) over the samples of a real ViT-B/224 x 32 capture?"""
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
        o = _m.raw_out.float()
        if g.dim() == 3:      # linear: (b, T, N)
            row = (g * g).sum(-1).reshape(-1)
        elif g.dim() == 4 and not hasattr(_m, "weight"):   # matmul (b, H, M, N): per image
            row = (g * g).sum(dim=(1, 2, 3))
        elif g.dim() == 4:    # conv (b, oc, H, W): per (image, pixel) -- the rows of its im2col GEMM
            row = (g * g).sum(1).reshape(-1)
        else:
            row = (g * g).reshape(g.shape[0], -1).sum(-1)
        srt = torch.sort(row, descending=True).values
        cum = torch.cumsum(srt, 0) / srt.sum()
        k8 = max(1, row.numel() // 8)
        first = row[:k8].sum() / row.sum()
        seen[_n] = (row.numel(), float(cum[k8 - 1]), float(first), float(row.max() / row.mean()))
        _m.calibrated = True
        for a in ("raw_input", "raw_out", "raw_grad"):
            delattr(_m, a)
    m.calibration_step2 = rec
with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
    HessianQuantCalibrator(net, wrapped, L(), sequential=False, batch_size=4).batching_quant_calib()
print("module: rows, mass in the top 1/8 rows, mass in the FIRST 1/8 rows, max/mean row mass")
for n, v in list(seen.items())[:14] + list(seen.items())[-8:]:
    print(f"{n:28s} {v[0]:6d}  top {v[1]:.3f}  first {v[2]:.3f}  max/mean {v[3]:.1f}")
