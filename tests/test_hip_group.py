"""GPU: p4v_calibrate_group -- the calibration_step2 of several modules in ONE call, their kernel launches grouped -- gives,
bit for bit, what the single-module calls give.

A grouped kernel (k_x_g, csrc/p4v_kernels.h) runs every member's own body on its own parameter block and scratch; the host side
(csrc/p4v_api.hip, Group) only merges the members' operation queues.  So nothing may differ -- not an interval, not a split --
whatever the mixture of module kinds, however the members diverge (pass-memo hits, empty survivor stages, a member that fails).

Reference: the loop these calls replace, utils/quant_calib.py:371-372 (sequential=False: modules are independent, :316-372).
"""
import contextlib
import io

import pytest
import torch
import torch.nn.functional as F

from tests.test_hip_production_path import _intervals, vit_like_grad

pytestmark = pytest.mark.gpu

PTQ4VIT = dict(metric="hessian", eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=3)


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from ptq4vit_amd import engine
    return engine


def _linear_kw(g, b, T, K, N, n_V, postgelu=False, metric="hessian", bits=8):
    w = torch.randn(N, K, generator=g) * 0.05 * torch.linspace(0.5, 1.5, N).view(-1, 1)
    bias = torch.randn(N, generator=g) * 0.1
    x = torch.randn(b, T, K, generator=g)
    if postgelu:
        x = F.gelu(x * 2)
    out = F.linear(x, w, bias)
    grad = vit_like_grad(out.shape, 1, g)
    hp = dict(PTQ4VIT, metric=metric)
    return dict(weight=w.cuda(), bias=bias.cuda(), x=x.cuda(), out=out.cuda(), grad=grad.cuda() if metric == "hessian" else None,
                w_bit=bits, a_bit=bits, n_V=n_V, n_H=1, n_a=1, postgelu=postgelu, **hp)


def _matmul_kw(g, b, H, M, K, N, sos=False):
    A = torch.randn(b, H, M, K, generator=g)
    if sos:
        A = torch.softmax(A * 3, dim=-1)
    B = torch.randn(b, H, K, N, generator=g)
    out = A @ B
    grad = vit_like_grad(out.shape, 2, g)
    return dict(A=A.cuda(), B=B.cuda(), out=out.cuda(), grad=grad.cuda(), A_bit=8, B_bit=8, sos=sos, **PTQ4VIT)


def _conv_kw(g):
    w = torch.nn.init.trunc_normal_(torch.empty(96, 3, 16, 16), std=0.02, generator=g)
    b = torch.randn(96, generator=g) * 0.02
    x = torch.randn(8, 3, 96, 96, generator=g)
    out = F.conv2d(x, w, b, stride=16)
    grad = torch.randn(out.shape, generator=g) * 1e-10
    return dict(weight=w.cuda(), bias=b.cuda(), x=x.cuda(), out=out.cuda(), grad=grad.cuda(), stride=(16, 16), padding=(0, 0),
                dilation=(1, 1), w_bit=8, a_bit=32, channelwise=True, **PTQ4VIT)


def _mixture(eng):
    g = torch.Generator().manual_seed(5)
    specs = []
    for i in range(3):                                                  # three same-shaped layers: the launches group
        specs.append(("linear", _linear_kw(g, 8, 197, 384, 384, 1)))
    specs.append(("linear", _linear_kw(g, 8, 197, 384, 1152, 3)))                            # qkv: three score blocks
    specs.append(("linear", _linear_kw(g, 8, 197, 1536, 384, 1, postgelu=True)))             # fc2: twin, K >= 1024
    specs.append(("linear", _linear_kw(g, 8, 197, 384, 1536, 1, bits=6)))                    # W6A6
    specs.append(("linear", _linear_kw(g, 8, 50, 96, 192, 1, metric="cosine")))              # never pruned, swapped sweep
    specs.append(("linear", _linear_kw(g, 8, 1, 384, 100, 1)))                               # head: one row per image
    for i in range(2):
        specs.append(("matmul", _matmul_kw(g, 8, 6, 197, 64, 197)))                          # q.k^T
    specs.append(("matmul", _matmul_kw(g, 8, 6, 197, 197, 64, sos=True)))                    # attn.v, split-of-softmax
    specs.append(("conv", _conv_kw(g)))
    return specs


def _jobs(eng, specs):
    return [getattr(eng, kind + "_job")(**kw) for kind, kw in specs]


def test_group_call_equals_the_single_calls_on_a_mixture_of_module_kinds(eng):
    specs = _mixture(eng)
    single = [eng.run_job(j) for j in _jobs(eng, specs)]
    torch.cuda.synchronize()
    eng.launch_counters(reset=True)
    grouped = eng.calibrate_group(_jobs(eng, specs))
    torch.cuda.synchronize()
    cnt = eng.launch_counters(reset=True)
    n = 0
    for (kind, _), a, b in zip(specs, single, grouped):
        for x, y in zip(a.outputs, b.outputs):
            if x is None:
                assert y is None
                continue
            assert torch.equal(x, y), f"{kind}: single {x.flatten()[:4].tolist()} vs grouped {y.flatten()[:4].tolist()}"
            n += x.numel()
    assert cnt["groups"] == 1 and cnt["issued"] < cnt["asked"], cnt
    print(f"[group] {len(specs)} members, {n} interval scalars bit-identical; launches asked for {cnt['asked']}, issued {cnt['issued']} "
          f"in {cnt['rounds']} rounds")


def test_group_of_one_and_empty_group(eng):
    g = torch.Generator().manual_seed(9)
    kw = _linear_kw(g, 4, 197, 192, 192, 1)
    a = eng.run_job(eng.linear_job(**kw))
    b = eng.calibrate_group([eng.linear_job(**kw)])[0]
    torch.cuda.synchronize()
    assert all(torch.equal(x, y) for x, y in zip(a.outputs, b.outputs))
    assert eng.calibrate_group([]) == []


def test_a_failing_member_fails_the_call_and_leaves_the_others_complete(eng):
    """One member whose workspace is too small (the engine's own P4V_ERR_WORKSPACE, raised while the group is running): the
    call raises the member's error; the other members are not left half-issued (their intervals equal the single calls')."""
    g = torch.Generator().manual_seed(11)
    good = _linear_kw(g, 4, 197, 192, 192, 1)
    ref = eng.run_job(eng.linear_job(**good))
    torch.cuda.synchronize()
    eng.release_workspace()
    jobs = [eng.linear_job(**good), eng.linear_job(**good), eng.linear_job(**good)]
    jobs[1].need = 4096                                 # (engine.workspace rounds up to ~1 MB: the planes alone need 5 MB)
    with pytest.raises(RuntimeError, match="workspace too small"):
        eng.calibrate_group(jobs)
    torch.cuda.synchronize()
    eng.release_workspace()
    for j in (jobs[0], jobs[2]):
        assert all(torch.equal(x, y) for x, y in zip(ref.outputs, j.outputs))


def test_group_under_the_engines_cross_check_and_without_pruning(eng):
    """The members under variant 134217728 (every pruned pass followed by the full sweep of the same pass; a differing selection
    is an error) and with the pruning off: same intervals -- the grouped launches carry device-side candidate ranges, per-block
    ranges and mapped-host read-backs of every member separately."""
    specs = _mixture(eng)[:6]
    base = eng.calibrate_group(_jobs(eng, specs))
    torch.cuda.synchronize()
    try:
        eng.debug_variant(134217728)
        chk = eng.calibrate_group(_jobs(eng, specs))
        eng.debug_variant(4194304)
        off = eng.calibrate_group(_jobs(eng, specs))
        torch.cuda.synchronize()
    finally:
        eng.debug_variant(0)
    for a, b, c in zip(base, chk, off):
        for x, y, z in zip(a.outputs, b.outputs, c.outputs):
            assert torch.equal(x, y) and torch.equal(x, z)


@pytest.mark.parametrize("model,calib", [("vit_base_patch16_224", 32), ("deit_tiny_patch16_224", 16)], ids=["vit-b-x32", "deit-tiny-x16"])
def test_whole_network_grouped_search_equals_the_per_module_search(eng, model, calib):
    """The calibrator's two search paths on the same network and images: one p4v_calibrate_group call over all modules (default),
    two concurrent group calls, and the per-module calls on four streams (P4V_GROUPED=0, rounds 2-5): every interval scalar
    bit-identical, and the grouped path issues a fraction of the launches."""
    from ptq4vit_amd.configs import PTQ4ViT
    from ptq4vit_amd.utils import models, net_wrap
    from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator
    torch.cuda.empty_cache()
    eng.release_workspace()
    net = models.get_net(model, seed=0, device="cuda")
    with contextlib.redirect_stdout(io.StringIO()):
        wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
    images = torch.randn(calib, 3, 224, 224, generator=torch.Generator().manual_seed(0)).cuda()

    class Loader:
        batch_size = calib

        def __iter__(self):
            yield images, None

    def calibrate(grouped, calls=1):
        for m in wrapped.values():
            m.mode = "raw"
        eng.launch_counters(reset=True)
        cal = HessianQuantCalibrator(net, wrapped, Loader(), sequential=False, batch_size=4)
        cal.search_grouped, cal.group_calls = grouped, calls
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            cal.batching_quant_calib()
        torch.cuda.synchronize()
        return _intervals(wrapped), eng.launch_counters(reset=True)

    one, c_one = calibrate(True, 1)
    two, c_two = calibrate(True, 2)
    per, c_per = calibrate(False)
    n = 0
    for name in one:
        for a, b, c in zip(one[name], two[name], per[name]):
            assert torch.equal(a, c), f"{name}: grouped {a.flatten()[:4].tolist()} vs per module {c.flatten()[:4].tolist()}"
            assert torch.equal(a, b), f"{name}: one group call vs two"
            n += a.numel()
    assert c_per["groups"] == 0 and c_per["issued"] == c_per["asked"], c_per
    assert c_one["groups"] == 1 and c_one["issued"] * 3 < c_per["issued"], (c_one, c_per)
    print(f"[group] {model} x {calib}: {n} interval scalars bit-identical; kernel launches per calibration: per module {c_per['issued']}, "
          f"one group {c_one['issued']} ({c_one['rounds']} rounds), two groups {c_two['issued']}")


def test_group_scratch_is_one_arena_within_its_budget(eng, monkeypatch):
    """The scratch of a group call is ONE arena per stream, carved by offset, and `_search_grouped` sizes the concurrent calls' arenas to
    a budget (P4V_GROUP_GIB, never more than what is free next to the resident caches): with a budget that holds only a few members a
    call runs as several p4v_calibrate_group batches REUSING the arena -- same intervals as the default, and the engine keeps no more
    scratch than the budget.  (Round 6: a buffer per member slot, each grown to the largest member it ever saw, ran Swin-B/384 x 128 out
    of memory next to its 193 GiB of captured tensors; reference utils/quant_calib.py:371-372 is the loop this replaces.)"""
    from ptq4vit_amd.configs import PTQ4ViT
    from ptq4vit_amd.utils import models, net_wrap
    from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator
    torch.cuda.empty_cache()
    eng.release_workspace()
    net = models.get_net("deit_tiny_patch16_224", seed=0, device="cuda")
    with contextlib.redirect_stdout(io.StringIO()):
        wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
    images = torch.randn(8, 3, 224, 224, generator=torch.Generator().manual_seed(1)).cuda()

    class Loader:
        batch_size = 8

        def __iter__(self):
            yield images, None

    def calibrate():
        for m in wrapped.values():
            m.mode = "raw"
        eng.launch_counters(reset=True)
        cal = HessianQuantCalibrator(net, wrapped, Loader(), sequential=False, batch_size=4)
        cal.search_grouped, cal.group_calls = True, 2
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            cal.batching_quant_calib()
        torch.cuda.synchronize()
        return _intervals(wrapped), eng.launch_counters(reset=True)

    ref, c_ref = calibrate()
    big = eng.workspace_bytes()
    assert c_ref["groups"] == 2, c_ref
    eng.release_workspace()
    budget = max(big // 8, 48 << 20)
    monkeypatch.setenv("P4V_GROUP_GIB", repr(budget / 2**30))
    small, c_small = calibrate()
    assert c_small["groups"] > c_ref["groups"], (c_small, c_ref)          # the calls ran as several batches
    for name in ref:
        for a, b in zip(ref[name], small[name]):
            assert torch.equal(a, b), name
    # two arenas (one per concurrent call), each within its half of the budget unless a single member needs more
    assert eng.workspace_bytes() < big, (eng.workspace_bytes(), big)
    print(f"[group] scratch kept by the engine: {big >> 20} MiB at the default budget, {eng.workspace_bytes() >> 20} MiB at {budget >> 20} MiB "
          f"({c_small['groups']} group calls instead of {c_ref['groups']})")
