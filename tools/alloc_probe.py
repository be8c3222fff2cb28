"""How fast does the device allocator hand out ~190 GiB: one block vs 450 blocks (capture caches of Swin-B/384 x 128)."""
import time
import torch
torch.cuda.init()
torch.empty(1, device="cuda")
GiB = 1 << 30
for label, sizes in (("1 x 150 GiB", [150 * GiB]), ("450 x 0.33 GiB", [int(0.3333 * GiB)] * 450), ("45 x 3.3 GiB", [int(3.333 * GiB)] * 45)):
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    t = time.time()
    bufs = [torch.empty(s, dtype=torch.uint8, device="cuda") for s in sizes]
    torch.cuda.synchronize()
    dt = time.time() - t
    t = time.time()
    for b in bufs[:3]:
        b[:: 1 << 21].fill_(1)          # touch one byte per 2 MiB page of the first blocks
    torch.cuda.synchronize()
    dt2 = time.time() - t
    b = None
    del bufs
    t = time.time(); torch.cuda.empty_cache(); torch.cuda.synchronize(); dt3 = time.time() - t
    print(f"{label}: alloc {dt:.2f} s, first touch of 3 blocks {dt2:.3f} s, free {dt3:.2f} s", flush=True)
