"""GPU: the drop-in module classes and the calibrator end to end.

* reference-captured tensors (tests/golden/minivit_ptq4vit.npz) -> module.calibration_step2() on the GPU
  -> the reference's calibrated intervals;
* the full HessianQuantCalibrator on the GPU (capture + search) checked module by module against the numpy
  oracle run on the very tensors the GPU capture produced;
* size-independent properties at the BASELINE shapes (ViT-B/224, 32 images): run-to-run determinism,
  agreement of the three sweep kernels, invariance of the selection under a power-of-two rescale of raw_grad.
"""
import json

import numpy as np
import pytest
import torch

from tests.helpers import CAPTURE_TOL, assert_on_candidate_grid, assert_scores_close, candidate_grid, grid_steps_between

pytestmark = pytest.mark.gpu


def _interval_parity(m, name, attr, got, want, ref_scores=None):
    """Bit-identical, or -- a near-tie resolved differently by a different summation order -- another entry of the SAME
    candidate table (exact fp32 ratio); returns (intervals, intervals that moved).  The split-of-softmax split and the
    A_interval derived from it must be equal."""
    got = torch.as_tensor(got).detach().cpu().numpy().reshape(-1)
    want = np.asarray(want).reshape(-1)
    assert got.shape == want.shape, (name, attr)
    if attr == "split" or (attr == "A_interval" and getattr(m, "_sos", False)):
        return want.size, int((got != want).sum())
    moved = assert_on_candidate_grid(got, want, candidate_grid(m.eq_alpha, m.eq_beta, m.eq_n), f"{name}.{attr}", ref_scores=ref_scores)
    return want.size, moved


def _mini():
    from ptq4vit_amd.configs import PTQ4ViT
    from ptq4vit_amd.utils import models, net_wrap
    g = np.load("tests/golden/minivit_ptq4vit.npz", allow_pickle=False)
    kw = json.loads(str(g["model_kwargs"]))
    net = models.get_net("vit_tiny_patch16_224", seed=0, device="cuda", **kw)
    wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
    return g, net, wrapped


def test_modules_reproduce_reference_intervals_from_reference_captures():
    g, net, wrapped = _mini()
    moved = total = 0
    for n, m in wrapped.items():
        key = n.replace(".", "__")
        t = lambda a: torch.from_numpy(g[f"{key}::{a}"]).cuda()
        m.raw_input = [t("A"), t("B")] if f"{key}::A" in g.files else t("x")
        m.raw_out, m.raw_grad = t("out"), t("grad")
        m.calibration_step2()
        assert m.calibrated
        for a in ("w_interval", "a_interval", "A_interval", "B_interval", "split"):
            if f"{key}::{a}" not in g.files:
                continue
            k, mv = _interval_parity(m, n, a, getattr(m, a), g[f"{key}::{a}"])
            total += k
            moved += mv
    print(f"[parity] mini ViT from the reference's captures: {total - moved}/{total} intervals bit-identical to the reference, "
          f"{moved} on another entry of the same candidate table (near-ties)")
    assert moved <= 0.03 * total


def _calibrate_and_compare_with_oracle(net, wrapped, images, min_exact=0.95, flat=()):
    """Run the calibrator on the GPU, record what every module captured, replay the oracle on those tensors."""
    from oracle.ptq4vit_oracle import ConvOracle, LinearOracle, MatMulOracle
    from ptq4vit_amd.quant_layers.conv import MinMaxQuantConv2d
    from ptq4vit_amd.quant_layers.linear import MinMaxQuantLinear
    from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator

    class Loader:
        batch_size = images.shape[0]

        def __iter__(self):
            yield images, torch.zeros(images.shape[0], dtype=torch.long)

    caps = {}
    for n, m in wrapped.items():
        orig = m.calibration_step2

        def rec(_o=orig, _m=m, _n=n):
            ri = _m.raw_input
            caps[_n] = ([x.cpu().numpy() for x in ri] if isinstance(ri, list) else ri.cpu().numpy(),
                        _m.raw_out.cpu().numpy(), _m.raw_grad.cpu().numpy())
            return _o()
        m.calibration_step2 = rec
    cal = HessianQuantCalibrator(net, wrapped, Loader(), sequential=False, batch_size=4)
    cal.batching_quant_calib()
    npy = lambda t: None if t is None else t.detach().cpu().numpy()
    moved = total = 0
    moved_names = []
    for n, m in wrapped.items():
        ri, ro, rg = caps[n]
        hp = dict(metric=m.metric, eq_alpha=m.eq_alpha, eq_beta=m.eq_beta, eq_n=m.eq_n, search_round=m.search_round)
        if isinstance(m, MinMaxQuantLinear):
            o = LinearOracle(npy(m.weight), npy(m.bias), w_bit=8, a_bit=8, n_V=m.n_V,
                             postgelu=type(m).__name__.startswith("PostGelu"), **hp)
            res = o.calibration_step2(ri, ro, rg)
        elif isinstance(m, MinMaxQuantConv2d):
            o = ConvOracle(npy(m.weight), npy(m.bias), stride=m.stride, w_bit=8, a_bit=32,
                           channelwise=type(m).__name__.startswith("Channelwise"), **hp)
            res = o.calibration_step2(ri, ro, rg)
            res.pop("a_interval")
        else:
            o = MatMulOracle(A_bit=8, B_bit=8, sos=type(m).__name__.startswith("SoS"), **hp)
            res = o.calibration_step2(ri[0], ri[1], ro, rg)
        # the oracle's LAST table of each operand (a selection further than one entry from the oracle's must be a tie by it)
        last = {}
        for tag, tab in o.trace:
            last[tag[0]] = tab
        tabs = {"w_interval": last.get("w"), "a_interval": last.get("a"), "A_interval": last.get("A"), "B_interval": last.get("B")}
        for a, want in res.items():
            k, mv = _interval_parity(m, n, a, getattr(m, a), want, ref_scores=tabs.get(a))
            if n in flat:
                continue      # counted separately below: a handful of samples and a flat metric tie to the last bit
            total += k
            moved += mv
            if mv:
                moved_names.append(f"{n}.{a}")
    print(f"[parity] calibrator vs oracle on the GPU captures: {total - moved}/{total} intervals bit-identical, {moved} on another "
          f"entry of the same candidate table {moved_names}; modules with flat metrics (any grid entry accepted): {list(flat)}")
    assert moved <= (1.0 - min_exact) * total, f"{moved}/{total} intervals differ from the oracle"
    with torch.no_grad():
        assert torch.isfinite(net(images)).all()      # every module now runs in quant_forward mode


def test_calibrator_end_to_end_vs_oracle_on_gpu_captures():
    g, net, wrapped = _mini()
    _calibrate_and_compare_with_oracle(net, wrapped, torch.from_numpy(g["images"]).cuda())


def test_baseptq_cosine_calibration_end_to_end_vs_oracle():
    """BASELINE.json config 0 in small: the BasePTQ config (cosine metric, one round, layer-wise conv, plain
    matmuls) through the whole calibrator, every module against the oracle."""
    import contextlib, io
    from ptq4vit_amd.configs import BasePTQ
    from ptq4vit_amd.utils import models, net_wrap
    net = models.get_net("deit_tiny_patch16_224", seed=5, device="cuda", img_size=32, patch_size=8, embed_dim=48, depth=2,
                         num_heads=3, num_classes=10)
    with contextlib.redirect_stdout(io.StringIO()):
        wrapped = net_wrap.wrap_modules_in_net(net, BasePTQ)
    images = torch.randn(4, 3, 32, 32, generator=torch.Generator().manual_seed(6)).cuda()
    _calibrate_and_compare_with_oracle(net, wrapped, images, min_exact=0.9, flat=("head",))   # head: 4 samples x 10 logits, cosine


def test_deit_tiny_224_baseptq_4_images_vs_the_reference_itself():
    """BASELINE.json config 0 at full size against the REFERENCE's own run of it (tests/golden/deit_tiny_224_baseptq_4img.npz,
    oracle/gen_golden.py::gen_deit_tiny: reference net_wrap + HessianQuantCalibrator.batching_quant_calib on the CPU, configs/
    BasePTQ.py as shipped -- cosine, so nothing depends on the rounding-noise raw_grad; reference utils/quant_calib.py:300-378,
    configs/BasePTQ.py:13-62).  Same seeded weights and images (checksums), raw logits to RAW_TOL.  Then for all 74 modules
    every calibrated interval is an entry of the reference's candidate table (exact fp32 ratio; the table itself may sit on an
    initial interval that differs in its last bits: the capture here is this GPU's fp32 GEMMs, the reference's was the CPU's)
    and the candidate it stands for is the reference's argmax or a NEAR-TIE BY THE REFERENCE'S OWN SCORE TABLE (stored in the
    fixture; TIE: the cosine tables of this run are flat to fp32 resolution around their maximum -- ~30 of 100 candidates lie
    within 1e-6 of it -- so the reference's own argmax is decided by rounding).  Counts printed.  The quantised logits are
    compared at the size of the quantisation error itself: an interval one grid step away re-draws the rounding pattern of a
    whole activation tensor.  A second calibration reproduces the first bit for bit."""
    import contextlib, io
    from ptq4vit_amd.configs import BasePTQ
    from ptq4vit_amd.quant_layers.conv import MinMaxQuantConv2d
    from ptq4vit_amd.quant_layers.linear import MinMaxQuantLinear
    from ptq4vit_amd.utils import models, net_wrap
    from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator
    RAW_TOL, TIE = 2e-5, 1e-5          # raw logits: fraction of the logit range; TIE: relative score gap by the reference's table
    g = np.load("tests/golden/deit_tiny_224_baseptq_4img.npz", allow_pickle=False)
    net = models.get_net("deit_tiny_patch16_224", seed=0, device="cuda")
    images = torch.randn(4, 3, 224, 224, generator=torch.Generator().manual_seed(0))
    assert abs(images.double().sum().item() - float(g["images_sum"])) <= 1e-9 * float(g["images_abs_sum"])
    w_sum = sum(p.double().abs().sum().item() for p in net.parameters())
    assert abs(w_sum - float(g["weights_abs_sum"])) <= 1e-9 * float(g["weights_abs_sum"]), "not the weights the reference calibrated"
    images = images.cuda()
    rng = float(g["raw_logits"].max() - g["raw_logits"].min())
    with torch.no_grad():
        raw = net(images).cpu().numpy()
    raw_err = np.abs(raw - g["raw_logits"]).max() / rng
    with contextlib.redirect_stdout(io.StringIO()):
        wrapped = net_wrap.wrap_modules_in_net(net, BasePTQ)
    assert list(wrapped) == [str(n) for n in g["names"]]           # 74 modules, the reference's wrapping order

    class Loader:
        batch_size = 4

        def __iter__(self):
            yield images, torch.zeros(4, dtype=torch.long)

    runs = []
    for _ in range(2):
        for m in wrapped.values():
            m.mode = "raw"
        with contextlib.redirect_stdout(io.StringIO()):
            HessianQuantCalibrator(net, wrapped, Loader(), sequential=False, batch_size=4).batching_quant_calib()
        runs.append({n: [getattr(m, a).clone() for a in ("w_interval", "a_interval", "A_interval", "B_interval") if hasattr(m, a)]
                     for n, m in wrapped.items()})
    for n in runs[0]:
        for a, b in zip(runs[0][n], runs[1][n]):
            assert torch.equal(a, b), n
    total = moved = rounded = 0
    dist, worst_gap = [], 0.0
    for n, m in wrapped.items():
        key = n.replace(".", "__")
        grid = candidate_grid(m.eq_alpha, m.eq_beta, m.eq_n)
        # which score table of the module's step 2 decides which interval (BasePTQ: one round)
        if isinstance(m, MinMaxQuantLinear):
            tabs = {"w_interval": 0, "a_interval": 1}
        elif isinstance(m, MinMaxQuantConv2d):
            tabs = {"w_interval": 0}
        else:
            tabs = {"A_interval": 0, "B_interval": 1}
        for a in ("w_interval", "a_interval", "A_interval", "B_interval"):
            if f"{key}::{a}" not in g.files:
                continue
            want = g[f"{key}::{a}"].reshape(-1)
            got = torch.as_tensor(getattr(m, a)).detach().cpu().numpy().reshape(-1)
            assert got.shape == want.shape, (n, a)
            if a not in tabs:                                       # the conv's a_interval: a_bit = 32, min-max of the images
                np.testing.assert_array_equal(got, want)
                continue
            ref_tab = g[f"{key}::scores_{tabs[a]}"].astype(np.float64).reshape(m.eq_n, -1)
            assert ref_tab.shape[1] == want.size, (n, a, ref_tab.shape)
            assert_on_candidate_grid(got, want, grid, f"{n}.{a}", tol=CAPTURE_TOL, ref_scores=ref_tab, tie_rtol=TIE)
            total += want.size
            for j, (x, y) in enumerate(zip(got, want)):
                if x == y:
                    continue
                steps = grid_steps_between(x, y, grid, tol=CAPTURE_TOL)
                if steps == 0:
                    rounded += 1
                    continue
                moved += 1
                dist.append(steps)
                ref_idx = int(np.argmax(ref_tab[:, j]))
                mine = int(np.argmin(np.abs(grid[:-1].astype(np.float64) / float(grid[ref_idx]) - float(x) / float(y))))
                gap = (ref_tab[ref_idx, j] - ref_tab[mine, j]) / abs(ref_tab[ref_idx, j])
                worst_gap = max(worst_gap, gap)
                assert gap <= TIE, f"{n}.{a}[{j}]: candidate {mine} vs the reference's {ref_idx}: not a tie by the reference's scores ({gap:.2e})"
    with torch.no_grad():
        q = net(images).cpu().numpy()
    q_err = np.abs(q - g["quant_logits"]).max() / rng
    quant_noise = np.abs(g["quant_logits"] - g["raw_logits"]).max() / rng
    print(f"[parity] DeiT-tiny/224 BasePTQ x4 vs the reference's own run: {total - moved - rounded}/{total} intervals bit-identical, "
          f"{rounded} the same candidate of a table that differs in its last bits, {moved} another candidate (grid steps away: "
          f"{sorted(dist)}; worst score gap by the reference's own tables {worst_gap:.1e}); raw logits {raw_err:.2e}, quantised "
          f"logits {q_err:.2e} of the logit range (quantisation error itself: {quant_noise:.2e}); argmax agreement "
          f"{int((q.argmax(1) == g['quant_logits'].argmax(1)).sum())}/4")
    assert raw_err <= RAW_TOL, raw_err
    assert moved <= 0.15 * total, (moved, total)
    assert q_err <= (1e-4 if moved == 0 else 2.0 * quant_noise), (q_err, quant_noise)


def test_swin_calibration_end_to_end_vs_oracle():
    """Swin: window attention (batch = images x windows, shifted windows with mask), bias-free `reduction` Linear."""
    import contextlib, io
    from ptq4vit_amd.configs import PTQ4ViT
    from ptq4vit_amd.utils import models, net_wrap
    net = models.get_net("swin_tiny_patch4_window7_224", seed=2, device="cuda", img_size=56, embed_dim=24, depths=(2, 2),
                         num_heads=(2, 4), num_classes=10)
    with contextlib.redirect_stdout(io.StringIO()):
        wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
    images = torch.randn(8, 3, 56, 56, generator=torch.Generator().manual_seed(4)).cuda()
    _calibrate_and_compare_with_oracle(net, wrapped, images)


# ---- BASELINE-size properties (ViT-B/224 qkv: 32 x 197 x 768 -> 2304, n_V = 3) ------------------------
@pytest.fixture(scope="module")
def vitb_qkv():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(32, 197, 768, generator=g).cuda()
    w = (torch.randn(2304, 768, generator=g) * 0.02).cuda()
    b = (torch.randn(2304, generator=g) * 0.02).cuda()
    out = torch.nn.functional.linear(x, w, b)
    grad = (torch.randn(out.shape, generator=g) * 1e-10).cuda()     # the magnitude the reference's KL gradient has
    return dict(weight=w, bias=b, x=x, out=out, grad=grad)


HP = dict(w_bit=8, a_bit=8, metric="hessian", eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=3, n_V=3, n_H=1, n_a=1)


def test_full_size_determinism_and_kernel_agreement(vitb_qkv):
    from ptq4vit_amd import engine
    r1 = engine.linear_calibrate(**vitb_qkv, want_scores=True, **HP)
    r2 = engine.linear_calibrate(**vitb_qkv, want_scores=True, **HP)
    for a, b in zip(r1, r2):
        assert torch.equal(a, b), "run-to-run results differ"
    engine.debug_variant(4)              # variant 4: stationary-operand sweep off -> streaming k_sweep2
    try:
        r3 = engine.linear_calibrate(**vitb_qkv, want_scores=True, **HP)
    finally:
        engine.debug_variant(0)
    assert_scores_close(r1[2].cpu().numpy(), r3[2].cpu().numpy(), rtol=2e-5, what="k_sweep3 vs k_sweep2")
    assert torch.equal(r1[3], r3[3]) and torch.equal(r1[0], r3[0]) and torch.equal(r1[1], r3[1])
    assert torch.isfinite(r1[2]).all() and (r1[2][:, 0] < 0).all() and (r1[2][:, 1, :, 0] < 0).all()


def test_full_size_selection_invariant_under_grad_rescale(vitb_qkv):
    """hessian score = -sum (g*d)^2: scaling raw_grad by 2^k scales every score by exactly 4^k."""
    from ptq4vit_amd import engine
    base = engine.linear_calibrate(**vitb_qkv, want_scores=True, **HP)
    scaled = dict(vitb_qkv, grad=vitb_qkv["grad"] * 1024.0)
    r = engine.linear_calibrate(**scaled, want_scores=True, **HP)
    assert torch.equal(base[3], r[3]) and torch.equal(base[0], r[0]) and torch.equal(base[1], r[1])
    assert torch.equal(base[2] * (1024.0 ** 2), r[2])


# ---- layer sharding on the GPU box: 2 ranks (gloo, same GPU) must reproduce the 1-rank intervals bit for bit ----
def _sharded_worker(rank, world, port, q):
    import os
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    g, net, wrapped = _mini()
    images = torch.from_numpy(g["images"]).cuda()

    class Loader:
        batch_size = images.shape[0]

        def __iter__(self):
            yield images, torch.zeros(images.shape[0], dtype=torch.long)

    # twice: the second calibration of a network replays the capture from the HIP graph recorded with hooks on EVERY module,
    # including those whose cache attributes the interval exchange deleted on this rank (the bench's warm-up + timed steps)
    for _ in range(2):
        for m in wrapped.values():
            m.mode = "raw"
        cal = HessianQuantCalibrator(net, wrapped, Loader(), sequential=False, batch_size=4)
        cal.batching_quant_calib()
    assert net.__dict__.get("_p4v_capture_graphs"), "the second calibration did not go through the capture graph"
    out = {}
    for n, m in wrapped.items():
        for a in ("w_interval", "a_interval", "A_interval", "B_interval", "split"):
            v = getattr(m, a, None)
            if v is not None:
                out[f"{n}.{a}"] = torch.as_tensor(v).detach().cpu().numpy().copy()
    q.put((rank, cal.timings["owned"], out))
    if world > 1:
        dist.destroy_process_group()


def test_sharded_calibration_matches_single_rank():
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    results = {}
    for world in (1, 2):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        q = ctx.Queue()
        procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=300) for _ in procs]
        for p in procs:
            p.join(60)
        results[world] = sorted(res, key=lambda t: t[0])
    single = results[1][0][2]
    owned = [r[1] for r in results[2]]
    assert sum(owned) == 14 and min(owned) >= 3, owned
    for rank, _, out in results[2]:
        assert out.keys() == single.keys()
        for k in single:
            np.testing.assert_array_equal(out[k], single[k], err_msg=f"rank {rank}: {k}")


def test_pass_memoisation_is_exact(vitb_qkv):
    """Rounds whose input interval was already evaluated are restored from the memo instead of recomputed: the
    calibrated intervals must be bit-identical to the full computation, and some passes must actually be skipped."""
    from ptq4vit_amd import engine
    engine.stats_reset()
    memo = engine.linear_calibrate(**vitb_qkv, **HP)                      # memoised (default)
    st = engine.stats_get()
    full = engine.linear_calibrate(**vitb_qkv, memoize=False, **HP)       # all 6 passes computed
    ref = engine.linear_calibrate(**vitb_qkv, want_scores=True, **HP)     # score tables requested -> never memoised
    torch.cuda.synchronize()
    assert torch.equal(memo[0], full[0]) and torch.equal(memo[1], full[1])
    assert torch.equal(memo[0], ref[0]) and torch.equal(memo[1], ref[1])
    assert st["memo_hits"] + st["memo_misses"] == 6 and st["memo_misses"] >= 2


@pytest.mark.parametrize("model,side,kw", [
    ("vit_tiny_patch16_224", 224, {}),
    ("swin_tiny_patch4_window7_224", 56, dict(img_size=56, embed_dim=24, depths=(2, 2), num_heads=(2, 4), num_classes=10)),
], ids=["vit", "swin"])
def test_graph_capture_equals_eager_capture(model, side, kw):
    """The HIP-graph replay of the sub-batch passes records exactly what the eager passes record."""
    import contextlib, io
    from ptq4vit_amd.configs import PTQ4ViT
    from ptq4vit_amd.utils import models, net_wrap
    from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator

    class L:
        def __init__(self, x):
            self.x, self.batch_size = x, x.shape[0]

        def __iter__(self):
            yield self.x, None

    dev = torch.device("cuda:0")
    net = models.get_net(model, seed=3, device=dev, **kw)
    with contextlib.redirect_stdout(io.StringIO()):
        wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
    x = torch.randn(8, 3, side, side, generator=torch.Generator().manual_seed(5)).to(dev)
    caps = []
    for use_graph in (True, False):
        cal = HessianQuantCalibrator(net, wrapped, L(x), sequential=False, batch_size=2, capture_batch_size=2)
        cal.use_graph = use_graph
        sm = cal._raw_pred_softmax()
        cal._capture(list(wrapped), sm, True)
        torch.cuda.synchronize()
        cap = {}
        for n, m in wrapped.items():
            ri = m.raw_input if isinstance(m.raw_input, list) else [m.raw_input]
            cap[n] = [t.clone() for t in ri] + [m.raw_out.clone(), m.raw_grad.clone()]
        caps.append(cap)
    for n in caps[0]:
        for a, b in zip(caps[0][n], caps[1][n]):
            assert a.shape == b.shape and torch.equal(a, b), n


def test_fresh_networks_replay_the_graph_of_their_architecture_with_their_own_weights(monkeypatch):
    """Round 6: the sub-batch passes of a FRESH network object replay the HIP graph recorded for its architecture on a private copy
    (utils/quant_calib.py, _arch_shadow) after its parameters have been copied into the graph's storage.  Three networks of one
    architecture with DIFFERENT weights: the third one's captured tensors -- taken through the architecture's graph, which was
    recorded with the second one's weights -- are bit-identical to its own eager capture."""
    import contextlib, io
    from ptq4vit_amd.configs import PTQ4ViT
    from ptq4vit_amd.utils import models, net_wrap
    from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator

    class L:
        def __init__(self, x):
            self.x, self.batch_size = x, x.shape[0]

        def __iter__(self):
            yield self.x, None

    dev = torch.device("cuda:0")
    x = torch.randn(8, 3, 64, 64, generator=torch.Generator().manual_seed(5)).to(dev)
    kw = dict(img_size=64, patch_size=16, embed_dim=96, depth=2, num_heads=3, num_classes=10)
    HessianQuantCalibrator._ARCH.clear()

    def capture(seed, arch_graphs):
        monkeypatch.setenv("P4V_ARCH_GRAPHS", "1" if arch_graphs else "0")
        net = models.get_net("vit_tiny_patch16_224", seed=seed, device=dev, **kw)
        with contextlib.redirect_stdout(io.StringIO()):
            wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
        cal = HessianQuantCalibrator(net, wrapped, L(x), sequential=False, batch_size=2)
        sm = cal._raw_pred_softmax()
        cal._capture(list(wrapped), sm, True)
        torch.cuda.synchronize()
        cap = {}
        for n, m in wrapped.items():
            ri = m.raw_input if isinstance(m.raw_input, list) else [m.raw_input]
            cap[n] = [t.clone() for t in ri] + [m.raw_out.clone(), m.raw_grad.clone()]
        return cap, bool(net.__dict__.get("_p4v_capture_graphs")) and net.__dict__.get("_p4v_capture_shadow") is None

    capture(1, True)                                  # first sighting of the architecture: eager
    assert not any("net" in r and r["net"] is not None for r in HessianQuantCalibrator._ARCH.values())
    capture(2, True)                                  # second sighting: the shadow and its graph are built (weights of seed 2)
    assert any(r.get("lanes") for r in HessianQuantCalibrator._ARCH.values())
    got, own = capture(3, True)                       # a third network, other weights, through the architecture's graph
    assert not own                                    # (no graph of its own was recorded)
    want, _ = capture(3, False)                       # its own eager passes
    for n in want:
        for a, b in zip(got[n], want[n]):
            assert a.shape == b.shape and torch.equal(a, b), n
    HessianQuantCalibrator._ARCH.clear()


def test_int8_quant_forward_matches_fake_quant_forward():
    """Row f-2: quant_forward on the int8 MFMA path (integer accumulation, scales in the epilogue) against the
    reference's fake-quant fp32 formulation, for every wrapped module of the calibrated mini ViT.
    Tolerance: 2e-5 of the output range (fp32 accumulation noise of the fp32 GEMM; the integer path is exact)."""
    g, net, wrapped = _mini()
    from ptq4vit_amd.quant_layers.conv import MinMaxQuantConv2d
    from ptq4vit_amd.quant_layers.matmul import MinMaxQuantMatMul
    checked = 0
    for n, m in wrapped.items():
        key = n.replace(".", "__")
        for a in ("w_interval", "a_interval", "A_interval", "B_interval", "split"):
            if f"{key}::{a}" in g.files:
                setattr(m, a, torch.from_numpy(g[f"{key}::{a}"]).cuda())
        m.calibrated = True
        if isinstance(m, MinMaxQuantConv2d):
            continue
        if isinstance(m, MinMaxQuantMatMul):
            ins = (torch.from_numpy(g[f"{key}::A"]).cuda(), torch.from_numpy(g[f"{key}::B"]).cuda())
            m._get_padding_parameters(*ins)
        else:
            ins = (torch.from_numpy(g[f"{key}::x"]).cuda(),)
        with torch.no_grad():
            m.int8_forward = True
            y_int = m.quant_forward(*ins)
            m.int8_forward = False
            y_ref = m.quant_forward(*ins)
        assert y_int.shape == y_ref.shape, n
        err = (y_int - y_ref).abs().max().item()
        assert err <= 2e-5 * y_ref.abs().max().item() + 1e-7, f"{n}: {err:.3e} vs range {y_ref.abs().max().item():.3e}"
        checked += 1
    assert checked == 13


def test_repeated_calibration_does_not_grow_memory():
    """Scratch buffers are per (device, stream): the calibrator must reuse its side streams, not strand a buffer set
    per calibration."""
    from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator
    g, net, wrapped = _mini()
    images = torch.from_numpy(g["images"]).cuda()

    class Loader:
        batch_size = images.shape[0]

        def __iter__(self):
            yield images, None

    used = []
    for _ in range(4):
        for m in wrapped.values():
            m.mode = "raw"
        HessianQuantCalibrator(net, wrapped, Loader(), sequential=False, batch_size=4).batching_quant_calib()
        torch.cuda.synchronize()
        used.append(torch.cuda.memory_allocated())
    assert used[3] <= used[1] + (1 << 20), used


@pytest.mark.parametrize("mode", ["sequential", "ragged_cpu_loader", "quant_calibrator_nograd", "hessian_quant_calib",
                                  "forward_mode_sequential", "forward_mode_parallel"])
def test_calibrator_variants_run(mode):
    """Less-travelled entry points of the reference's calibrators (quant_calib.py:28-93, 95-171, 216-298) run through
    the GPU engine: sequential capture (predecessors quantised), a calibration set that is not a multiple of the
    sub-batch and lives on the host, the gradient-free QuantCalibrator.batching_quant_calib, the non-batching
    HessianQuantCalibrator.quant_calib."""
    import contextlib, io
    from ptq4vit_amd.configs import BasePTQ, PTQ4ViT
    from ptq4vit_amd.utils import models, net_wrap
    from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator, QuantCalibrator
    cfg = BasePTQ if mode == "quant_calibrator_nograd" else PTQ4ViT
    net = models.get_net("vit_tiny_patch16_224", seed=1, device="cuda", img_size=32, patch_size=8, embed_dim=48, depth=2,
                         num_heads=3, num_classes=10)
    if mode in ("hessian_quant_calib", "forward_mode_sequential", "forward_mode_parallel"):
        # quant_calib() feeds calibration_step2(x): the non-batching classes (reference linear.py:94-347, matmul.py:75-388)
        from ptq4vit_amd.quant_layers.linear import PostGeluPTQSLQuantLinear, PTQSLQuantLinear
        from ptq4vit_amd.quant_layers.matmul import PTQSLQuantMatMul, SoSPTQSLQuantMatMul

        class cfg:  # noqa: N801
            hp = dict(metric="hessian" if mode == "hessian_quant_calib" else "L2_norm", search_round=2, eq_alpha=0.01,
                      eq_beta=1.2, eq_n=100)

            @staticmethod
            def get_module(kind, *a, **k):
                if kind == "qlinear_MLP_2":
                    return PostGeluPTQSLQuantLinear(*a, **k, **cfg.hp)
                if kind.startswith("qlinear"):
                    return PTQSLQuantLinear(*a, **k, n_V=3 if kind == "qlinear_qkv" else 1, **cfg.hp)
                return (SoSPTQSLQuantMatMul if kind == "qmatmul_scorev" else PTQSLQuantMatMul)(**cfg.hp)

        with contextlib.redirect_stdout(io.StringIO()):
            wrapped = net_wrap.wrap_certain_modules_in_net(net, cfg, [0, 1], ["qkv", "proj", "fc1", "fc2", "matmul1", "matmul2", "head"])
        assert len(wrapped) == 13
    else:
        with contextlib.redirect_stdout(io.StringIO()):
            wrapped = net_wrap.wrap_modules_in_net(net, cfg)
    n_img = 10 if mode == "ragged_cpu_loader" else 8
    images = torch.randn(n_img, 3, 32, 32, generator=torch.Generator().manual_seed(2))
    if mode != "ragged_cpu_loader":
        images = images.cuda()

    class Loader:
        batch_size = n_img

        def __iter__(self):
            yield images, torch.zeros(n_img, dtype=torch.long)

    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        if mode == "sequential":
            HessianQuantCalibrator(net, wrapped, Loader(), sequential=True, batch_size=4).batching_quant_calib()
        elif mode == "ragged_cpu_loader":
            HessianQuantCalibrator(net, wrapped, Loader(), sequential=False, batch_size=4).batching_quant_calib()
        elif mode == "quant_calibrator_nograd":
            QuantCalibrator(net, wrapped, Loader(), sequential=False).batching_quant_calib()
        elif mode == "hessian_quant_calib":
            HessianQuantCalibrator(net, wrapped, Loader(), sequential=False, batch_size=4).quant_calib()
        else:   # forward-mode calibration: modes calibration_step1 / calibration_step2 (reference quant_calib.py:28-93)
            QuantCalibrator(net, wrapped, Loader(), sequential=(mode == "forward_mode_sequential")).quant_calib()
    assert all(m.mode == "quant_forward" and m.calibrated for m in wrapped.values())
    with torch.no_grad():
        out = net(images.cuda())
    assert torch.isfinite(out).all()
    for m in wrapped.values():
        for a in ("w_interval", "a_interval", "A_interval", "B_interval"):
            v = getattr(m, a, None)
            if v is not None:
                v = v[0] if isinstance(v, (list, tuple)) else v
                assert torch.isfinite(torch.as_tensor(v)).all() and (torch.as_tensor(v) > 0).all()


def test_full_size_proj_layer_against_the_oracle():
    """BASELINE size, no reduction: ViT-B/224 `proj` (32 x 197 x 768 -> 768, W8A8, Hessian metric, 3 rounds x 100
    candidates, gradients of the magnitude the reference's KL loss produces) -- every score table of the HIP path
    against the numpy oracle, and the calibrated intervals (tie-aware)."""
    from oracle.ptq4vit_oracle import LinearOracle
    from ptq4vit_amd import engine
    from tests.helpers import assert_argmax_tie_aware, assert_scores_close
    g = torch.Generator().manual_seed(3)
    x = torch.randn(32, 197, 768, generator=g)
    w = torch.randn(768, 768, generator=g) * 0.02
    b = torch.randn(768, generator=g) * 0.02
    out = torch.nn.functional.linear(x, w, b)
    grad = torch.randn(out.shape, generator=g) * 1e-10
    hp = dict(w_bit=8, a_bit=8, metric="hessian", eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=3, n_V=1)
    o = LinearOracle(w.numpy(), b.numpy(), **hp)
    want = o.calibration_step2(x.numpy(), out.numpy(), grad.numpy())
    w_iv, a_iv, scores, best = engine.linear_calibrate(weight=w.cuda(), bias=b.cuda(), x=x.cuda(), out=out.cuda(),
                                                       grad=grad.cuda(), n_H=1, n_a=1, want_scores=True, **hp)
    torch.cuda.synchronize()
    scores, best = scores.cpu().numpy(), best.cpu().numpy()
    flips = 0
    for r in range(3):
        for k, (tab, idx) in enumerate(((scores[r, 0], best[r, 0]), (scores[r, 1][:, :1], best[r, 1][:1]))):
            ref = o.trace[2 * r + k][1].reshape(tab.shape)
            if flips == 0:      # after a tie flip the two searches continue from different intervals
                assert_scores_close(tab, ref, what=f"round {r} {'wa'[k]}")
                flips += int(not np.array_equal(idx, np.argmax(ref, axis=0)))
                assert_argmax_tie_aware(idx, ref, what=f"round {r} {'wa'[k]}")
    if flips == 0:
        np.testing.assert_array_equal(w_iv.cpu().numpy(), want["w_interval"].reshape(-1))
        np.testing.assert_array_equal(a_iv.cpu().numpy(), want["a_interval"].reshape(-1))


def _intervals(wrapped):
    out = {}
    for n, m in wrapped.items():
        out[n] = [torch.as_tensor(getattr(m, a)).detach().clone() for a in ("w_interval", "a_interval", "A_interval", "B_interval", "split")
                  if getattr(m, a, None) is not None and not isinstance(getattr(m, a), (list, tuple))]
    return out


def test_cached_capture_graph_groups_and_oom_replan_reproduce_the_intervals():
    """One network, calibrated four ways -- eager capture (first calibration), cached HIP graph (second calibration on),
    a cache budget that forces several capture groups (all served by the one cached graph), and an out-of-memory error
    in the middle of a group (budget halved, the rest re-planned) -- ends with bit-identical intervals."""
    from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator
    g, net, wrapped = _mini()
    images = torch.from_numpy(g["images"]).cuda()

    class Loader:
        batch_size = images.shape[0]

        def __iter__(self):
            yield images, None

    def calibrate(**attrs):
        for m in wrapped.values():
            m.mode = "raw"
        cal = HessianQuantCalibrator(net, wrapped, Loader(), sequential=False, batch_size=2, capture_batch_size=2)
        for k, v in attrs.items():
            setattr(cal, k, v)
        cal.batching_quant_calib()
        torch.cuda.synchronize()
        return cal, _intervals(wrapped)

    _, first = calibrate()
    assert not net.__dict__.get("_p4v_capture_graphs")                 # 4 sub-batches, first calibration: eager
    _, second = calibrate()
    graphs = net.__dict__["_p4v_capture_graphs"]
    assert len(graphs) == 1                                              # recorded on the second calibration ...
    entry = next(iter(graphs.values()))
    _, third = calibrate()
    assert next(iter(net.__dict__["_p4v_capture_graphs"].values())) is entry   # ... and replayed on the third
    sizes = HessianQuantCalibrator(net, wrapped, Loader(), batch_size=2)._estimate_cache_bytes(list(wrapped))
    cal4, fourth = calibrate(cache_budget_bytes=int(sum(sizes.values()) / 3.5))
    assert next(iter(net.__dict__["_p4v_capture_graphs"].values())) is entry

    # out of memory while the second module of a group is searched
    calls = {"n": 0}
    victim = list(wrapped)[1]
    orig = wrapped[victim].calibration_step2

    def boom():
        calls["n"] += 1
        if calls["n"] == 1:
            raise torch.cuda.OutOfMemoryError("simulated")
        return orig()
    wrapped[victim].calibration_step2 = boom
    try:
        _, fifth = calibrate(search_streams=1, cache_budget_bytes=int(sum(sizes.values()) * 2))
    finally:
        del wrapped[victim].calibration_step2
    assert calls["n"] == 2
    for other in (second, third, fourth, fifth):
        assert set(other) == set(first)
        for n in first:
            for a, b in zip(first[n], other[n]):
                assert torch.equal(a, b), n


def test_capture_pass_size_does_not_change_the_captured_tensors():
    """DeiT-tiny/224, 14 images (ragged last reference sub-batch): capture passes of 8 images (`capture_batch_size`,
    opt-in) against the reference's passes of batch_size = 4 (quant_calib.py:309-356) for a FIXED random target
    distribution (with the network's own prediction as the target, raw_grad is rounding noise in the reference as here):
    raw_input / raw_out / raw_grad agree to GEMM rounding."""
    import contextlib, io
    from ptq4vit_amd.configs import PTQ4ViT
    from ptq4vit_amd.utils import models, net_wrap
    from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator
    net = models.get_net("deit_tiny_patch16_224", seed=2, device="cuda")
    with contextlib.redirect_stdout(io.StringIO()):
        wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
    images = torch.randn(14, 3, 224, 224, generator=torch.Generator().manual_seed(4)).cuda()
    target = torch.softmax(torch.randn(14, 1000, generator=torch.Generator().manual_seed(5)), dim=-1).cuda()

    class Loader:
        batch_size = 14

        def __iter__(self):
            yield images, None

    caps = []
    for cbs in (None, 8):
        cal = HessianQuantCalibrator(net, wrapped, Loader(), sequential=False, batch_size=4, capture_batch_size=cbs)
        assert cal._capture_bs() == (8 if cbs else 4)
        cal._capture(list(wrapped), target, True)
        torch.cuda.synchronize()
        cap = {}
        for n, m in wrapped.items():
            ri = m.raw_input if isinstance(m.raw_input, list) else [m.raw_input]
            cap[n] = [t.clone() for t in ri] + [m.raw_out.clone(), m.raw_grad.clone()]
        caps.append(cap)
    worst = 0.0
    for n in caps[0]:
        for a, b in zip(caps[0][n], caps[1][n]):
            assert a.shape == b.shape and float(a.abs().max()) > 0
            worst = max(worst, float((a - b).abs().max() / a.abs().max()))
    print(f"[capture] passes of 8 vs 4 images, fixed target: captured tensors within {worst:.1e} of the tensor maximum")
    assert worst <= 5e-5, worst


def _rccl_worker(port, q):
    import os
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    from ptq4vit_amd.utils import shard
    from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        g, net, wrapped = _mini()
        images = torch.from_numpy(g["images"]).cuda()

        class Loader:
            batch_size = images.shape[0]

            def __iter__(self):
                yield images, None

        HessianQuantCalibrator(net, wrapped, Loader(), sequential=False, batch_size=4).batching_quant_calib()
        before = _intervals(wrapped)
        n = shard.exchange_intervals(wrapped, {name: 0 for name in wrapped})      # all_reduce + all_gather on device tensors
        dist.barrier()
        after = _intervals(wrapped)
        same = all(torch.equal(a, b) for k in before for a, b in zip(before[k], after[k]))
        q.put(("ok", int(n), same))
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001 - reported to the parent
        q.put(("error", repr(e), False))


def test_interval_exchange_runs_over_rccl():
    """The only collective of the default multi-GPU path (shard.exchange_intervals: one all_reduce of the slot table, one
    all_gather of the interval vector) on the `nccl` (= RCCL) backend with device tensors.  One GPU per box, so world_size is
    1 -- what this pins is that the RCCL code path (device placement, dtypes, the calls themselves) runs and is the identity."""
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(port, q))
    p.start()
    status, n, same = q.get(timeout=300)
    p.join(60)
    assert status == "ok", n
    assert n > 0 and same


def test_whole_calibration_is_run_to_run_deterministic():
    """DeiT-tiny/224, 8 images, 74 modules on three search streams: four calibrations of one network (eager capture, graph
    recording, graph replay twice) end with bit-identical intervals -- no result depends on how the streams interleave."""
    import contextlib, io
    from ptq4vit_amd.configs import PTQ4ViT
    from ptq4vit_amd.utils import models, net_wrap
    from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator
    net = models.get_net("deit_tiny_patch16_224", seed=9, device="cuda")
    with contextlib.redirect_stdout(io.StringIO()):
        wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
    images = torch.randn(8, 3, 224, 224, generator=torch.Generator().manual_seed(10)).cuda()

    class Loader:
        batch_size = 8

        def __iter__(self):
            yield images, None

    runs = []
    for _ in range(4):
        for m in wrapped.values():
            m.mode = "raw"
        with contextlib.redirect_stdout(io.StringIO()):
            HessianQuantCalibrator(net, wrapped, Loader(), sequential=False, batch_size=4).batching_quant_calib()
        torch.cuda.synchronize()
        runs.append(_intervals(wrapped))
    for other in runs[1:]:
        for n in runs[0]:
            for a, b in zip(runs[0][n], other[n]):
                assert torch.equal(a, b), n
