"""Layer sharding over the GPUs of one node + the final interval exchange.

With ``sequential=False`` every module's step 2 depends only on its own cached tensors, so modules are
independent units: longest-processing-time assignment by search FLOPs, no data-path collective, and ONE
all-gather of a fixed-layout fp32 interval vector at the end (a few KB: latency-bound, so xGMI bandwidth
does not matter).  The reference has no collective communication at all; this is the
multi-GPU design BASELINE.json asks for.
"""
import torch

from .intervals import INTERVAL_ATTRS, install_intervals      # the exchange installs what load_intervals installs

try:
    import torch.distributed as dist
except Exception:  # pragma: no cover
    dist = None


def rank_world():
    if dist is not None and dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def module_cost(module):
    """Relative search time of one module.  Linears: MACs per token of one candidate GEMM (K*N).  The patch
    embedding runs fp32-operand MFMA (~1/16 of the int8 rate).  Attention matmuls are small and epilogue-bound;
    their cost is set from the measured ViT-B breakdown (profiles/): about a proj-sized linear."""
    w = getattr(module, "weight", None)
    if w is None:
        return 0.6e6
    if w.dim() == 4:
        return float(w.numel()) * 4.0
    return float(w.numel())


def module_cost_ms(module, cache_bytes):
    """Predicted search time (ms, one MI355X, single stream) of one module from its captured size; constants fitted to
    the per-module measurements of a ViT-B/224 x 32 calibration (tools/module_times.py, round 2 kernels: qkv 4.2, proj 1.9,
    fc1 5.3, fc2 8.5, q.k 2.9, attn.v 2.4, patch embedding 7.6, head 0.6 ms for the three rounds with the pass memo).  Only
    the RATIOS matter (LPT balance)."""
    w = getattr(module, "weight", None)
    cls = type(module).__name__
    if w is None:                                       # matmul: cache = A + B + 2 * out (fp32)
        per_mac = 4.2e-9 if cls.startswith("SoS") else 3.9e-9
        # cache_bytes / 4 = Z (M K + K N + 2 M N); the sweeps cost ~ Z M N K; approximate with (cache / 4)^(3/2) / sqrt(Z)
        # being overkill, use the dominant square score matrix: out elements ~ cache / 16, K ~ 64
        t = 0.4 + per_mac * (cache_bytes / 16.0) * 64.0
        mn = getattr(module, "_p4v_out_mn", None)
        if mn is not None and not cls.startswith("SoS"):
            # the q.k^T sweeps are epilogue-bound on 128 x 128 tiles: what they cost follows the PADDED score matrix
            # (197 tokens: 1.69 x the valid area, the fit above; Swin windows of 144: 3.16 x -- measured 9.3 ms against 4.6)
            area = float(max(1, mn[0] * mn[1]))
            pad = (-(-mn[0] // 128) * 128) * (-(-mn[1] // 128) * 128) / area
            pad16 = (-(-mn[0] // 16) * 16) * (-(-mn[1] // 16) * 16) / area
            if pad / pad16 >= 1.8:                    # k_sweep9 (16 x 16 blocks): Swin windows, 6.1 ms measured at 144 tokens
                pad = 2.3 * pad16
            t = 0.4 + (t - 0.4) * pad / 1.689
        return t
    if w.dim() == 4:                                    # patch embedding: fp32-operand MFMA path
        k = w[0].numel()
        rows = cache_bytes / 4.0 / max(1.0, 2.0 * w.shape[0] + k)
        return 0.5 + 1.9e-9 * rows * k * w.shape[0]
    n_out, k = w.shape
    rows = cache_bytes / 4.0 / (k + 2.0 * n_out)        # cache = x + out + grad
    t = 0.78 + 1.93e-6 * k * n_out * rows / 6304.0
    return t * (1.6 if cls.startswith("PostGelu") else 1.0)


def assign_modules(wrapped_modules, world, costs=None):
    """LPT: heaviest module first onto the currently lightest rank.  Deterministic -> identical on every rank."""
    names = list(wrapped_modules)
    if costs is None:
        costs = {n: module_cost(wrapped_modules[n]) for n in names}
    order = sorted(names, key=lambda n: (-costs[n], names.index(n)))
    load = [0.0] * world
    owner = {}
    for n in order:
        r = min(range(world), key=lambda i: (load[i], i))
        owner[n] = r
        load[r] += costs[n]
    return owner


def _pack(module):
    vals, meta = [], []
    for a in INTERVAL_ATTRS:
        v = getattr(module, a, None)
        if v is None:
            continue
        if isinstance(v, (list, tuple)):  # non-batching post-GELU class keeps [tensor, float]
            v = v[0]
        t = torch.as_tensor(v, dtype=torch.float32).detach().reshape(-1)
        vals.append(t)
        meta.append((a, tuple(torch.as_tensor(v).shape)))
    return vals, meta


_HDR = 9 * len(INTERVAL_ATTRS)        # per module: for each attribute (1 + ndim, up to 8 dims), as exactly representable floats


def _slot_capacity(module):
    """Interval scalars a module can produce at most -- from attributes EVERY rank knows before any search (no communication):
    Linear n_V * n_H + n_a, Conv2d one per output channel + 1, MatMul one per head and operand + the split (the head count is
    only known to the owner, after its capture: 128 heads unless the module carries `_p4v_interval_slots`; exceeding a slot raises)."""
    w = getattr(module, "weight", None)
    if w is None:
        return int(getattr(module, "_p4v_interval_slots", 2 * 128 + 2))
    if w.dim() == 4:
        return int(w.shape[0]) + 2
    return int(getattr(module, "n_V", 1)) * int(getattr(module, "n_H", 1)) + int(getattr(module, "n_a", 1)) + 2


def exchange_intervals(wrapped_modules, owner, install_own=False):
    """All ranks end up with every module's calibrated intervals: ONE all_gather of a fixed-layout fp32 vector (BASELINE.json's
    north_star: "RCCL over xGMI only for the final scale gather").  The layout -- per module a header with the attributes'
    shapes, then `_slot_capacity(module)` value slots -- follows from the module list alone, so no rank has to be told
    anything before the gather (rounds 2-5 ran an all_reduce of the slot table first).  `install_own`: also install this rank's
    own modules from the gathered buffer (tools/measure_exchange.py: with one rank there is nobody else's module to install, and
    installing is most of the host-side cost)."""
    rank, world = rank_world()
    names = list(wrapped_modules)
    dev = torch.device("cuda", torch.cuda.current_device()) if (torch.cuda.is_available() and dist.get_backend() == "nccl") else torch.device("cpu")
    base, total = {}, 0
    for n in names:
        base[n] = total
        total += _HDR + _slot_capacity(wrapped_modules[n])
    import numpy as np
    host = np.zeros(total, dtype=np.float32)  # (assembled on the host, ONE tensor each way: element-wise tensor indexing is slow)
    pieces, where = [], []
    for n in names:
        if owner[n] != rank:
            continue
        vals, meta = _pack(wrapped_modules[n])
        off, used = base[n] + _HDR, 0
        for (a, shp), v in zip(meta, vals):
            if len(shp) > 8:
                raise ValueError(f"{n}.{a}: interval tensors of more than 8 dimensions are not exchanged")
            h = base[n] + 9 * INTERVAL_ATTRS.index(a)
            host[h] = float(1 + len(shp))
            for k, d in enumerate(shp):
                host[h + 1 + k] = float(int(d))
            if used + v.numel() > _slot_capacity(wrapped_modules[n]):
                raise ValueError(f"{n}: {used + v.numel()} interval scalars exceed the exchange slot of {_slot_capacity(wrapped_modules[n])}")
            pieces.append(v.to(dev).reshape(-1))
            where.append((off + used, v.numel()))
            used += v.numel()
    vec = torch.from_numpy(host).to(dev)
    if pieces:
        idx = torch.cat([torch.arange(o, o + k) for o, k in where]).to(dev)
        vec.index_copy_(0, idx, torch.cat(pieces))
    parts = [torch.empty(total, dtype=torch.float32, device=dev) for _ in range(world)]
    dist.all_gather(parts, vec)        # THE collective: ~100 KB per rank (RCCL over xGMI on the GPU box, gloo in tests)
    gathered = torch.stack(parts)
    hdr = gathered.cpu() if gathered.device.type != "cpu" else gathered      # headers are read on the host (a few thousand ints)
    for n in names:
        r = owner[n]
        if r == rank and not install_own:
            continue
        m = wrapped_modules[n]
        mdev = next((p.device for p in m.parameters()), dev) if hasattr(m, "parameters") else dev
        row = hdr[r, base[n]: base[n] + _HDR].tolist()
        vals, off = {}, base[n] + _HDR
        for j, a in enumerate(INTERVAL_ATTRS):
            nd = int(row[9 * j])
            if nd == 0:
                continue
            shp = tuple(int(d) for d in row[9 * j + 1: 9 * j + nd])
            numel = 1
            for d in shp:
                numel *= d
            vals[a] = gathered[r, off: off + numel].reshape(shp)
            off += numel
        # the state the owner's calibration_step2 left behind (utils/intervals.py); views of `gathered`, which belongs to this
        # call alone: no copy kernel per attribute (74 modules x 2 attributes of tiny launches were most of the exchange's time)
        install_intervals(m, vals, mdev, clone=(gathered.device != torch.device(mdev)))
    return total


# ---- sub-batch sharded capture ---------------------------------------------------------------------------------
def _module_pieces(module, with_grad):
    """The per-sub-batch tensor lists a capture leaves on a module, in a fixed order: inputs, out, (grad)."""
    ri = module.raw_input
    lists = list(ri) if (isinstance(ri, list) and ri and isinstance(ri[0], list)) else [ri]
    lists = lists + [module.raw_out]
    if with_grad:
        lists.append(module.raw_grad)
    return lists


def exchange_captures(wrapped_modules, owner, n_sub, grad_names):
    """Every rank ran the capture passes of the sub-batches i with i % world == rank, hooking ALL modules; afterwards
    each module's pieces travel to its owner in ONE `all_to_all_single` (RCCL over xGMI: every GPU sends to its seven
    peers over their own links at the same time; gloo in the tests).  Send buffer of a rank = for each destination, that
    owner's modules x this rank's sub-batches, padded to a common slot count so that every sender uses the same layout.  The
    owner reassembles the full tensors in sub-batch order -- bit-identical to a capture of all sub-batches on one GPU (each
    sub-batch pass is the same computation wherever it runs).  Non-owners drop their pieces.

    Traffic per GPU: (its modules' cache) x (world - 1) / world in and about as much out -- for ViT-B/224 x 32 images on
    8 GPUs about 1 GB each way, against 7/8 of the forward/backward passes saved.
    """
    rank, world = rank_world()
    names = list(wrapped_modules)
    slots = -(-n_sub // world)                       # sub-batches per rank, padded
    data_dev = None
    for n in names:
        lists = _module_pieces(wrapped_modules[n], n in grad_names)
        if lists[0]:
            data_dev = lists[0][0].device
            break
    # RCCL moves device buffers; gloo (CPU tests, or several ranks sharing one GPU) needs host buffers
    backend_dev = data_dev if dist.get_backend() == "nccl" else torch.device("cpu")
    layouts, sizes, chunks = [], [], []              # per destination: [(name, list index, piece shape)], elements
    for dst in range(world):
        layout, numel = [], 0
        for n in names:
            if owner[n] != dst:
                continue
            for li, lst in enumerate(_module_pieces(wrapped_modules[n], n in grad_names)):
                shp = tuple(lst[0].shape)
                layout.append((n, li, shp))
                for k in range(slots):
                    if k < len(lst):
                        chunks.append(lst[k].reshape(-1).to(device=backend_dev, dtype=torch.float32))
                    else:
                        chunks.append(torch.zeros(lst[0].numel(), dtype=torch.float32, device=backend_dev))
                numel += slots * lst[0].numel()
        layouts.append(layout)
        sizes.append(numel)
    send = torch.cat(chunks) if chunks else torch.zeros(0, dtype=torch.float32, device=backend_dev)
    del chunks
    for n in names:                                   # the pieces are in `send` now; what this rank owns comes back below
        m = wrapped_modules[n]
        m.raw_input = m.raw_out = None
        if n in grad_names:
            m.raw_grad = None
    per_src = sizes[rank]                             # every sender's block for this rank has this rank's layout
    recv = torch.empty(per_src * world, dtype=torch.float32, device=backend_dev)
    dist.all_to_all_single(recv, send, output_split_sizes=[per_src] * world, input_split_sizes=sizes)
    del send
    # reassembly: piece (sender, slot) of every tensor goes to its sub-batch position.  On a GPU all of them in ONE launch
    # (p4v_multi_copy; a copy per piece is ~2000 launches for ViT-B on 8 ranks -- as long as the search of a rank's modules)
    on_gpu = data_dev is not None and data_dev.type == "cuda"
    if on_gpu and recv.device != data_dev:
        recv = recv.to(data_dev)                     # (gloo with several ranks on one GPU: the collective ran on host buffers)
    full, off, rows = {}, 0, []
    for (n, li, shp) in layouts[rank]:
        per = 1
        for d in shp:
            per *= d
        t = torch.empty((n_sub * shp[0],) + shp[1:], dtype=torch.float32, device=data_dev)
        for src in range(world):
            base = src * per_src + off
            for k in range(slots):
                i = src + k * world                   # global sub-batch index of (sender, slot)
                if i >= n_sub:
                    continue
                if on_gpu:
                    rows.append([recv.data_ptr() + 4 * (base + k * per), t.data_ptr() + 4 * i * per, 4 * per])
                else:
                    t[i * shp[0]:(i + 1) * shp[0]].copy_(recv[base + k * per: base + (k + 1) * per].reshape(shp))
        off += slots * per
        full[(n, li)] = t
    if rows:
        from .. import engine
        table = torch.tensor(rows, dtype=torch.int64).to(data_dev)
        engine.multi_copy(table, len(rows), 0, max(r[2] for r in rows), data_dev)
        torch.cuda.current_stream(data_dev).synchronize()      # `recv` and `table` are released below
    del recv
    for n in names:
        if owner[n] != rank:
            continue
        m = wrapped_modules[n]
        with_g = n in grad_names
        nl = sum(1 for (nn, _, _) in layouts[rank] if nn == n)
        ts = [full[(n, li)] for li in range(nl)]
        n_in = nl - 1 - (1 if with_g else 0)
        m.raw_input = ts[0] if n_in == 1 else ts[:n_in]
        m.raw_out = ts[n_in]
        if with_g:
            m.raw_grad = ts[n_in + 1]


# ---- replicated or sharded capture: a rank-invariant cost model --------------------------------------------------
def capture_cost_ms(wrapped_modules, sizes):
    """Predicted time (ms, one MI355X) of ALL capture passes of a calibration, from rank-invariant quantities only: the
    MACs of the wrapped GEMMs (forward + the activation-gradient half of the backward; weight gradients are skipped) at the
    fp32 rate the sub-batch passes reach (21.5 TMAC/s: ViT-B/224 x 32 images = 2 x 560 GMAC in 52 ms, profiles/r2_bench.json).
    `sizes[name]` = bytes of the module's captured tensors (x + out + grad; matmuls A + B + out + grad)."""
    macs = 0.0
    for n, m in wrapped_modules.items():
        w = getattr(m, "weight", None)
        c = float(sizes.get(n, 0))
        if w is None:
            macs += (c / 16.0) * 64.0                 # out elements ~ cache / 16, K ~ head dim
        elif w.dim() == 4:
            k = w[0].numel()
            macs += c / 4.0 / max(1.0, 2.0 * w.shape[0] + k) * k * w.shape[0]
        else:
            macs += c / 4.0 / (w.shape[1] + 2.0 * w.shape[0]) * w.shape[0] * w.shape[1]
    return 2.0 * macs / 21.5e9


def choose_capture_mode(wrapped_modules, sizes, world, n_sub, gb_per_s_per_peer=None):
    """"sharded" when running 1/world of the capture passes per rank and moving the pieces to the module owners is predicted
    to beat every rank running all passes, else "replicated" -- the DEFAULT, BASELINE.json's plan (no data-path collective).
    `gb_per_s_per_peer`: the all_to_all_single rate per peer MEASURED on this process group (`a2a_rate_gbps()`: a 4 MB-per-peer
    exchange at the first calibration); None = not measured = "replicated" (rounds 3-5 assumed 40 GB/s and switched to the
    GB-scale collective at >= 4 ranks on that guess).  Every input is the same on every rank (module list, captured sizes from
    the shape probe, world size, the rate agreed by all_reduce), so all ranks take the same branch -- required: the sharded path
    is a collective.  A GPU receives from its world - 1 peers over as many xGMI links at once, and pays two extra passes over its
    share in HBM for packing and reassembly."""
    if world < 2 or n_sub < 2 or not gb_per_s_per_peer or gb_per_s_per_peer <= 0:
        return "replicated"
    t_cap = capture_cost_ms(wrapped_modules, sizes)
    share = sum(float(v) for v in sizes.values()) / world              # bytes a rank ends up owning (LPT: about equal)
    moved = share * (world - 1) / world
    t_xfer = moved / (gb_per_s_per_peer * 1e6 * (world - 1)) + 3.0 * share / 2.0e9 + 2.0      # ms
    passes_saved = 1.0 - float(-(-n_sub // world)) / n_sub
    return "sharded" if t_cap * passes_saved > 1.5 * t_xfer + 5.0 else "replicated"


_A2A_RATE = None


def a2a_rate_gbps(mb_per_peer=4):
    """GB/s per peer of all_to_all_single on this process group, measured once (a `mb_per_peer` MB block to every peer, best of
    three after a warm-up) and AGREED between the ranks (all_reduce MIN: the slowest rank's figure, the same on all of them).
    0.0 when the collective is unavailable.  Only called when the sharded capture is a candidate (world >= 2, auto mode)."""
    global _A2A_RATE
    if _A2A_RATE is not None:
        return _A2A_RATE
    import time
    rank, world = rank_world()
    if world < 2 or not all_to_all_available():
        _A2A_RATE = 0.0
        return _A2A_RATE
    on_gpu = torch.cuda.is_available() and dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    n = mb_per_peer * (1 << 20) // 4
    send = torch.ones(world * n, dtype=torch.float32, device=dev)
    recv = torch.empty_like(send)
    best = float("inf")
    for i in range(4):
        if on_gpu:
            torch.cuda.synchronize()
        dist.barrier()
        t = time.perf_counter()
        dist.all_to_all_single(recv, send, output_split_sizes=[n] * world, input_split_sizes=[n] * world)
        if on_gpu:
            torch.cuda.synchronize()
        dt = time.perf_counter() - t
        if i > 0:
            best = min(best, dt)
    rate = torch.tensor([4.0 * n / best / 1e9], dtype=torch.float64, device=dev)
    dist.all_reduce(rate, op=dist.ReduceOp.MIN)
    _A2A_RATE = float(rate.item())
    return _A2A_RATE


_A2A_OK = None


def all_to_all_available():
    """One-time check that the process group's all_to_all_single works here, AGREED between the ranks (a tiny all_reduce(MIN)
    of the local outcomes -- the collective the interval exchange needs anyway): the sharded capture is only entered when
    every rank can run it; otherwise all of them stay with the replicated capture."""
    global _A2A_OK
    if _A2A_OK is not None:
        return _A2A_OK
    rank, world = rank_world()
    dev = torch.device("cuda", torch.cuda.current_device()) if (torch.cuda.is_available() and dist.get_backend() == "nccl") else torch.device("cpu")
    ok = 1
    try:
        send = torch.full((world * 4,), float(rank), dtype=torch.float32, device=dev)
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, output_split_sizes=[4] * world, input_split_sizes=[4] * world)
        want = torch.arange(world, dtype=torch.float32, device=dev).repeat_interleave(4)
        ok = int(bool(torch.equal(recv, want)))
    except Exception as e:  # pragma: no cover - depends on the backend build
        print(f"[ptq4vit_amd] all_to_all_single unavailable on rank {rank} ({type(e).__name__}: {e}); replicated capture")
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    _A2A_OK = bool(int(flag.item()))
    return _A2A_OK
