"""Timing probe of k_bound (stage B1) on ViT-B layer shapes: python tools/bound_probe.py [tuning value of key 12]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ptq4vit_amd import engine
tv = int(sys.argv[1]) if len(sys.argv) > 1 else 0
engine.debug_tuning(12, tv)
g = torch.Generator().manual_seed(0)
for name, K, N, nV in (("proj", 768, 768, 1), ("qkv", 768, 2304, 3), ("fc1", 768, 3072, 1)):
    x = torch.randn(32, 197, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) * 0.02).cuda()
    b = torch.zeros(N).cuda()
    out = torch.nn.functional.linear(x, w, b)
    grad = torch.randn(out.shape, generator=g).cuda() * 1e-10
    grad[:, 0] *= 300
    args = dict(weight=w, bias=b, x=x, out=out, grad=grad, w_bit=8, a_bit=8, metric="hessian", eq_alpha=0.01, eq_beta=1.2, eq_n=100,
                search_round=1, n_V=nV, n_H=1, n_a=1)
    engine.linear_calibrate(**args)
    engine.stats_reset(); engine.stats_enable(True)
    for _ in range(3):
        engine.linear_calibrate(**args)
    torch.cuda.synchronize(); engine.stats_get()
    recs = [r for r in engine.stats_launches() if r["kernel"] == "k_bound"]
    engine.stats_enable(False)
    print(name, "tuning", tv, "k_bound launches", len(recs), "avg us", round(sum(r["ms"] for r in recs) / max(1, len(recs)) * 1e3, 1), "min us", round(min(r["ms"] for r in recs) * 1e3, 1))
