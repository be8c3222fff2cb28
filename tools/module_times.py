"""Per-module search time of a ViT-B/224 x 32 calibration on one stream (fits shard.module_cost_ms)."""
import collections, contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ptq4vit_amd.configs import PTQ4ViT
from ptq4vit_amd.utils import models, net_wrap, shard
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

times = collections.defaultdict(list)
for rep in range(3):
    for n, m in wrapped.items():
        m.mode = "raw"
        m.__dict__.pop("calibration_step2", None)
        orig = m.calibration_step2
        def timed(_o=orig, _n=n):
            torch.cuda.synchronize(); t = time.time(); r = _o(); torch.cuda.synchronize()
            times[_n].append((time.time() - t) * 1e3); return r
        m.calibration_step2 = timed
    cal = HessianQuantCalibrator(net, wrapped, L(), sequential=False, batch_size=4)
    cal.search_streams = 1
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        cal.batching_quant_calib()
for m in wrapped.values():
    m.__dict__.pop("calibration_step2", None)
sizes = cal._estimate_cache_bytes(list(wrapped))
by = collections.defaultdict(lambda: [0.0, 0.0, 0])
for n, m in wrapped.items():
    key = n.split(".")[-1] if "blocks" in n or "layers" in n else n
    by[key][0] += min(times[n][1:]); by[key][1] += shard.module_cost_ms(m, sizes[n]); by[key][2] += 1
print(f"{model} x {n_img}: measured (ms per module, best of 2 warm runs) vs shard.module_cost_ms")
for k, (a, b, c) in by.items():
    print(f"  {k:12s} n={c:3d}  measured {a / c:7.2f}  model {b / c:7.2f}")
print("  total measured", round(sum(v[0] for v in by.values()), 1), "model", round(sum(v[1] for v in by.values()), 1))
