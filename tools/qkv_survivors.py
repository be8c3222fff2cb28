"""Experiment: per score block (q / k / v) survivors of the pruned qkv weight search of a ViT-B/224 x 32 calibration, as a
function of the slice size -- would per-block candidate ranges in stage B2 pay?  python tools/qkv_survivors.py [block]"""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ptq4vit_amd import engine
from ptq4vit_amd.configs import PTQ4ViT
from ptq4vit_amd.utils import models, net_wrap
from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator
blk = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
net = models.get_net("vit_base_patch16_224", seed=0, device=dev)
with contextlib.redirect_stdout(io.StringIO()):
    wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
images = torch.randn(32, 3, 224, 224, generator=torch.Generator().manual_seed(0)).to(dev)
class L:
    batch_size = 32
    def __iter__(self):
        yield images, None
caps = {}
for n, m in wrapped.items():
    if n.endswith("attn.qkv") or n.endswith("mlp.fc1"):
        orig = m.calibration_step2
        def rec(_o=orig, _m=m, _n=n):
            caps[_n] = (_m.raw_input.clone(), _m.raw_out.clone(), _m.raw_grad.clone())
            return _o()
        m.calibration_step2 = rec
with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
    HessianQuantCalibrator(net, wrapped, L(), sequential=False, batch_size=4).batching_quant_calib()
for name in (f"blocks.{blk}.attn.qkv", f"blocks.{blk}.mlp.fc1"):
    m = wrapped[name]
    x, out, grad = caps[name]
    nV = m.n_V
    w_iv, a_iv, scores, best = engine.linear_calibrate(weight=m.weight.data, bias=m.bias.data, x=x, out=out, grad=grad, w_bit=8, a_bit=8,
                                                       metric="hessian", eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=1, n_V=nV, n_H=1, n_a=1,
                                                       want_scores=True)
    tot = scores[0, 0].double()                     # (100, nV) totals of the weight search, round 1
    M = x.shape[0] * x.shape[1]
    N = out.shape[-1]
    g2 = (grad.double() ** 2).reshape(M, nV, N // nV)
    mass_rows = g2.sum(dim=(1, 2))
    order = torch.argsort(mass_rows, descending=True)
    mass_blk = g2.sum(-1)                           # (M, nV)
    # per-candidate per-row per-block error terms: recompute the candidates' outputs with fp32 fake quant (GPU, chunked)
    x2 = x.reshape(M, -1)
    a0 = x2.abs().max() / 127.5
    xq = torch.clamp(torch.round(x2 / a0), -128, 127) * a0
    wv = m.weight.data.view(nV, N // nV, -1)
    w0 = wv.abs().amax(dim=(1, 2)) / 127.5
    mult = torch.tensor([0.01 + i * (1.2 - 0.01) / 100 for i in range(100)], device=dev)
    o2, gr2 = out.reshape(M, N), grad.reshape(M, N)
    err = torch.empty(100, M, nV, dtype=torch.float64, device=dev)     # sum over the block's features of (g * delta)^2
    for c in range(100):
        s = (mult[c] * w0).view(nV, 1, 1)
        wq = (torch.clamp(torch.round(wv / s), -128, 127) * s).reshape(N, -1)
        d = (o2 - torch.nn.functional.linear(xq, wq, m.bias.data)) * gr2
        err[c] = (d.double() ** 2).reshape(M, nV, N // nV).sum(-1)
    total = err.sum(1)                              # (100, nV)
    Lstar = total.min(0).values                     # best total per block (scores are minus these, up to the normalisation)
    print(f"== {name}: M {M} N {N} n_V {nV}; engine argmax {best[0, 0].tolist()} vs recomputed {total.argmin(0).tolist()}")
    for k in (128, 256, 512, 1280, 2304, 3200):
        rows = order[:k]
        part = err[:, rows].sum(1)                  # (100, nV) partial sums over the slice (rows ranked by TOTAL mass, as the engine)
        surv = (part <= Lstar[None] * (1 + 1e-4))   # partial error not above the best total -> cannot be excluded
        share = (mass_blk[rows].sum(0) / mass_blk.sum(0)).tolist()
        hull = surv.any(1).nonzero().flatten()
        print(f"   slice {k:5d} rows: share of the weight per block {[round(v, 3) for v in share]}; survivors per block {surv.sum(0).tolist()}, "
              f"hull over the blocks {int(hull.max() - hull.min() + 1) if hull.numel() else 0}")
    # per-block ranking of the rows instead of one ranking
    for k in (512,):
        surv_b = []
        for j in range(nV):
            rows = torch.argsort(mass_blk[:, j], descending=True)[:k]
            surv_b.append(int((err[:, rows, j].sum(1) <= Lstar[j] * (1 + 1e-4)).sum()))
        print(f"   slice {k} rows ranked PER BLOCK: survivors per block {surv_b}")
