"""Where does the capture pass spend its time?  (tuning aid)  python tools/prof_capture.py"""
import os, sys, time, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import SyntheticLoader as Loader  # noqa
from ptq4vit_amd.utils import models, net_wrap
from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator
from ptq4vit_amd.configs import PTQ4ViT as cfg_mod


def main():
    dev = torch.device("cuda:0")
    net = models.get_net("vit_base_patch16_224", seed=0, device=dev)
    with contextlib.redirect_stdout(io.StringIO()):
        wrapped = net_wrap.wrap_modules_in_net(net, cfg_mod)
    g = torch.Generator().manual_seed(1)
    loader = Loader(torch.randn(32, 3, 224, 224, generator=g).to(dev))
    cal = HessianQuantCalibrator(net, wrapped, loader, sequential=False, batch_size=4)
    if os.environ.get("P4V_USE_GRAPH"):
        cal.use_graph = True
    names = list(wrapped)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        sm = cal._raw_pred_softmax()
        torch.cuda.synchronize(); t1 = time.time()
        cal._estimate_cache_bytes(names)
        torch.cuda.synchronize(); t2 = time.time()
        cal._capture(names, sm, True)
        t3 = time.time()
        torch.cuda.synchronize(); t4 = time.time()
        print(f"raw_pred {1e3*(t1-t0):.1f} ms, probe {1e3*(t2-t1):.1f} ms, capture enqueue {1e3*(t3-t2):.1f} ms, capture drain {1e3*(t4-t3):.1f} ms")
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        cal._capture(names, sm, True)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=18, max_name_column_width=60))


if __name__ == "__main__":
    main()
