"""GPU: where does the int8 quant_forward of a network leave the fp32 fake-quant forward of the same modules on the CPU?
DeiT-tiny/224 BasePTQ with the REFERENCE's intervals (tests/golden/deit_tiny_224_baseptq_4img.npz) on both devices, 4 seeded
images: every wrapped module's output on the GPU against the CPU's, (a) inside the network (differences accumulate) and (b) with
the CPU's own input fed to the GPU module (the module alone)."""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests.test_top1_agreement import _net_with_reference_intervals

torch.manual_seed(0)
x = torch.randn(4, 3, 224, 224, generator=torch.Generator().manual_seed(7))
net_c, wr_c = _net_with_reference_intervals("cpu")
net_g, wr_g = _net_with_reference_intervals("cuda")
cap_c, cap_g = {}, {}
def hook(store):
    def mk(name):
        def f(mod, inp, out):
            store[name] = ([t.detach().float().cpu().clone() for t in inp], out.detach().float().cpu().clone())
        return f
    return mk
hs = [m.register_forward_hook(hook(cap_c)(n)) for n, m in wr_c.items()] + [m.register_forward_hook(hook(cap_g)(n)) for n, m in wr_g.items()]
with torch.no_grad():
    yc = net_c(x)
    yg = net_g(x.cuda()).float().cpu()
for h in hs:
    h.remove()
print("logits: max |gpu - cpu| / range = %.3e" % (float((yg - yc).abs().max()) / float(yc.max() - yc.min())))
print("%-28s %-34s %12s %12s" % ("module", "class", "in-network", "module alone"))
for n, m in wr_g.items():
    ins_c, out_c = cap_c[n]
    _, out_g = cap_g[n]
    scale = float(out_c.abs().max()) + 1e-30
    e_net = float((out_g - out_c).abs().max()) / scale
    with torch.no_grad():
        o = m(*[t.cuda() for t in ins_c]).float().cpu()
    e_mod = float((o - out_c).abs().max()) / scale
    flag = "  <--" if e_mod > 1e-4 else ""
    print("%-28s %-34s %12.3e %12.3e%s" % (n, type(m).__name__, e_net, e_mod, flag))
