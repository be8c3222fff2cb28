"""What the final interval exchange of a sharded calibration costs (utils/shard.py::exchange_intervals: ONE all_gather of a
fixed-layout fp32 vector, round 6), measured where it can be: RCCL with one rank on the GPU box (the launch + synchronisation
floor of the collective; a one-GPU box cannot show xGMI), gloo with 2 ranks on the host (another transport: an upper bound for a
latency-bound message of ~100 KB).  74 calibrated modules of DeiT-tiny/224 (the reference's intervals).  With ONE rank nobody
else's module is installed -- and installing is most of the host-side cost -- so the one-rank run also times the exchange with
every module installed from the gathered buffer (`install_own`): at world W a rank installs (W - 1) / W of the modules, which is
what tools/predict_scale.py adds to the one-rank figure.

  python tools/measure_exchange.py --backend nccl            (GPU box)
  python tools/measure_exchange.py --backend gloo --world 2  (anywhere)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, backend, port, iters, out):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if backend == "nccl":
        torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from ptq4vit_amd.utils import shard
    from tests.test_top1_agreement import _net_with_reference_intervals
    net, wrapped = _net_with_reference_intervals("cuda" if backend == "nccl" else "cpu")
    names = list(wrapped)
    owner = {n: i % world for i, n in enumerate(names)}
    sync = (lambda: torch.cuda.synchronize()) if backend == "nccl" else (lambda: None)
    for _ in range(3):
        total = shard.exchange_intervals(wrapped, owner)
    sync(); dist.barrier()
    ts, ti = [], []
    for _ in range(iters):
        t0 = time.perf_counter()
        shard.exchange_intervals(wrapped, owner)
        sync()
        ts.append(time.perf_counter() - t0)
    for _ in range(iters):
        t0 = time.perf_counter()
        shard.exchange_intervals(wrapped, owner, install_own=True)      # every module installed from the gathered buffer
        sync()
        ti.append(time.perf_counter() - t0)
    # the collectives alone (the rest of exchange_intervals is host-side packing / installing of 74 modules)
    dev = torch.device("cuda", 0) if backend == "nccl" else torch.device("cpu")
    vec = torch.zeros(total, dtype=torch.float32, device=dev)
    parts = [torch.empty_like(vec) for _ in range(world)]
    cs = []
    for _ in range(iters):
        sync(); t0 = time.perf_counter()
        dist.all_gather(parts, vec)
        sync()
        cs.append(time.perf_counter() - t0)
    if rank == 0:
        ts.sort(); cs.sort(); ti.sort()
        ver = None
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
        except Exception:
            pass
        res = {"backend": backend, "world": world, "scalars": int(total), "modules": len(names), "iters": iters, "rccl_version": ver,
               "exchange_intervals_ms": {"median": 1e3 * ts[len(ts) // 2], "min": 1e3 * ts[0], "max": 1e3 * ts[-1]},
               "exchange_installing_every_module_ms": {"median": 1e3 * ti[len(ti) // 2], "min": 1e3 * ti[0], "max": 1e3 * ti[-1]},
               "collectives_only_ms": {"median": 1e3 * cs[len(cs) // 2], "min": 1e3 * cs[0], "max": 1e3 * cs[-1]}}
        print(json.dumps(res))
        if out:
            cur = {}
            if os.path.exists(out):
                try:
                    cur = json.load(open(out))
                except Exception:
                    cur = {}
            cur[f"{backend}_world{world}"] = res
            json.dump(cur, open(out, "w"), indent=1)
    dist.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--port", type=int, default=29731)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    if a.world == 1:
        worker(0, 1, a.backend, a.port, a.iters, a.out)
    else:
        import torch.multiprocessing as mp
        mp.spawn(worker, args=(a.world, a.backend, a.port, a.iters, a.out), nprocs=a.world, join=True)
