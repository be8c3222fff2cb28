"""Multi-GPU path on CPU: layer sharding (LPT) and the final interval all-gather, world_size 2, gloo."""
import os
import socket

import torch
import torch.multiprocessing as mp

from ptq4vit_amd.utils import shard


class FakeLinear(torch.nn.Module):
    def __init__(self, n_in, n_out, n_V):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.zeros(n_out, n_in))
        self.n_V = n_V
        self.w_interval = self.a_interval = None
        self.raw_input = self.raw_out = self.raw_grad = None


class FakeMatMul(torch.nn.Module):
    def __init__(self, heads, sos):
        super().__init__()
        self.heads, self.sos = heads, sos
        self.A_interval = self.B_interval = None
        if sos:
            self.split = None
        self.raw_input = self.raw_out = self.raw_grad = None


def _build():
    mods = {}
    for b in range(3):
        mods[f"blocks.{b}.attn.qkv"] = FakeLinear(48, 144, 3)
        mods[f"blocks.{b}.attn.matmul1"] = FakeMatMul(3, False)
        mods[f"blocks.{b}.attn.matmul2"] = FakeMatMul(3, True)
        mods[f"blocks.{b}.mlp.fc1"] = FakeLinear(48, 192, 1)
    mods["head"] = FakeLinear(48, 10, 1)
    return mods


def _calibrate(name, m):
    """Deterministic stand-in for calibration_step2: intervals are a function of the module name."""
    seed = sum(ord(ch) for ch in name)
    g = torch.Generator().manual_seed(seed)
    if isinstance(m, FakeLinear):
        m.w_interval = torch.rand(m.n_V, 1, 1, 1, generator=g)
        m.a_interval = torch.rand(1, 1, generator=g)
    else:
        m.B_interval = torch.rand(1, m.heads, 1, 1, 1, 1, 1, generator=g)
        if m.sos:
            m.split = torch.rand((), generator=g)
            m.A_interval = m.split / 127
        else:
            m.A_interval = torch.rand(1, m.heads, 1, 1, 1, 1, 1, generator=g)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mods = _build()
    owner = shard.assign_modules(mods, world)
    for n, m in mods.items():
        if owner[n] == rank:
            _calibrate(n, m)
    total = shard.exchange_intervals(mods, owner)
    ref = _build()
    ok = True
    for n, m in ref.items():
        _calibrate(n, m)
        for a in shard.INTERVAL_ATTRS:
            want, got = getattr(m, a, None), getattr(mods[n], a, None)
            if want is None:
                continue
            ok &= got is not None and tuple(got.shape) == tuple(want.shape) and torch.equal(got.float(), want.float())
    q.put((rank, ok, total, sorted(set(owner.values()))))
    dist.destroy_process_group()


def test_assignment_is_deterministic_and_balanced():
    mods = _build()
    o1, o2 = shard.assign_modules(mods, 4), shard.assign_modules(mods, 4)
    assert o1 == o2 and set(o1.values()) == {0, 1, 2, 3}
    load = [0.0] * 4
    for n, r in o1.items():
        load[r] += shard.module_cost(mods[n])
    assert max(load) <= 1.6 * (sum(load) / 4)


def test_interval_exchange_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(ok for _, ok, _, _ in res), res
    assert res[0][2] == res[1][2] > 0 and res[0][3] == [0, 1]


# ---- sub-batch sharded capture: the real calibrator on the mini ViT, CPU, 2 and 3 ranks ----------------------------
def _capture_worker(rank, world, port, q):
    import contextlib, io, json
    import numpy as np
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ptq4vit_amd.configs import PTQ4ViT
    from ptq4vit_amd.utils import models, net_wrap, quant_calib
    g = np.load("tests/golden/minivit_ptq4vit.npz", allow_pickle=False)
    kw = json.loads(str(g["model_kwargs"]))
    net = models.get_net("vit_tiny_patch16_224", seed=0, device="cpu", **kw)
    with contextlib.redirect_stdout(io.StringIO()):
        wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
    images = torch.from_numpy(g["images"])

    class Loader:
        batch_size = images.shape[0]

        def __iter__(self):
            yield images, None

    seen = {}
    for n, m in wrapped.items():
        def rec(_m=m, _n=n):     # stand-in for the GPU search: record what this rank would have calibrated from
            ri = _m.raw_input
            seen[_n] = ([t.clone() for t in ri] if isinstance(ri, list) else [ri.clone()]) + [_m.raw_out.clone(), _m.raw_grad.clone()]
            _m.calibrated = True
            _m.w_interval = torch.zeros(1)
        m.calibration_step2 = rec
    cal = quant_calib.HessianQuantCalibrator(net, wrapped, Loader(), sequential=False, batch_size=2)   # 4 sub-batches
    with contextlib.redirect_stdout(io.StringIO()):
        cal.batching_quant_calib()
    ok, nmine = True, 0
    for n, ts in seen.items():
        assert cal.owner[n] == rank
        nmine += 1
        key = n.replace(".", "__")
        want = ([g[f"{key}::A"], g[f"{key}::B"]] if f"{key}::A" in g.files else [g[f"{key}::x"]]) + [g[f"{key}::out"], g[f"{key}::grad"]]
        for t, w in zip(ts, want):
            ok &= tuple(t.shape) == tuple(w.shape)
    q.put((rank, ok, nmine, {n: [t.numpy() for t in ts] for n, ts in seen.items()}))
    dist.destroy_process_group()


def _run_capture(world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_capture_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    return res


def test_sharded_capture_reassembles_the_single_process_capture():
    """Each rank runs 1/world of the sub-batch passes for ALL modules; after the per-owner gather every owner holds
    exactly (bit for bit) the tensors a single process captures from all sub-batches."""
    import numpy as np
    single = _run_capture(1)[0]
    assert single[1] and single[2] == 14
    for world in (2, 3):
        res = _run_capture(world)
        assert all(ok for _, ok, _, _ in res)
        assert sum(nm for _, _, nm, _ in res) == 14
        merged = {}
        for _, _, _, d in res:
            merged.update(d)
        assert set(merged) == set(single[3])
        for n, ts in merged.items():
            for a, b in zip(ts, single[3][n]):
                np.testing.assert_array_equal(a, b, err_msg=n)
