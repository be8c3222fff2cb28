#!/usr/bin/env python
"""Benchmark of the PTQ4ViT calibration hot path on MI355X (contract: see the task statement / DESIGN.md).

A *step* is one full ``HessianQuantCalibrator(net, wrapped, loader, sequential=False, batch_size=4)
.batching_quant_calib()`` -- the region the reference times (example/test_all.py:31-34) -- on a ViT-B/224 with
seeded random weights and 32 seeded synthetic images that are resident in HBM before the clock starts.
value = wrapped modules calibrated per second, whole job (74 modules per step, sharded over the ranks for N > 1).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # before the first GPU call (what ptq4vit_amd.configure_runtime() does; the
                                                    # package import itself no longer touches the environment)

import numpy as np  # noqa: E402
import torch  # noqa: E402


class SyntheticLoader:
    """Calib-loader contract of the reference (utils/datasets.py:88-94): ONE batch of `num` images, `.batch_size`."""

    def __init__(self, images):
        self.images = images
        self.batch_size = images.shape[0]

    def __iter__(self):
        yield self.images, torch.zeros(self.images.shape[0], dtype=torch.long)


def search_macs(wrapped, shapes, calib, eq_n=100, rounds=3):
    """Algorithmic MACs of the reference's candidate-sweep GEMMs for one calibration, from the
    per-image operand shapes seen by a probe forward (`shapes[name]` = (input shapes, output shape) for 1 image)."""
    from ptq4vit_amd.quant_layers.conv import MinMaxQuantConv2d
    from ptq4vit_amd.quant_layers.linear import MinMaxQuantLinear
    lin = mm = conv = 0.0
    for name, m in wrapped.items():
        ins, out = shapes[name]
        r_, n_ = getattr(m, "search_round", rounds), getattr(m, "eq_n", eq_n)      # per module (BasePTQ: one round)
        if isinstance(m, MinMaxQuantLinear):
            rows = calib * int(torch.tensor(ins[0][:-1]).prod())
            lin += r_ * 2 * n_ * rows * m.in_features * m.out_features
        elif isinstance(m, MinMaxQuantConv2d):
            k = m.in_channels * m.kernel_size[0] * m.kernel_size[1]
            conv += r_ * n_ * calib * out[2] * out[3] * k * m.out_channels
        else:
            sos = type(m).__name__.startswith("SoS")
            A, B = ins
            per = calib * A[0] * A[1] * A[2] * A[3] * B[3]          # (windows x) heads x M x K x N per image
            mm += r_ * ((20 + n_) if sos else 2 * n_) * per
    return lin, mm, conv


def probe_shapes(net, wrapped, image):
    shapes, hooks = {}, []
    for n, m in wrapped.items():
        hooks.append(m.register_forward_hook(
            lambda mod, inp, out, _n=n: shapes.__setitem__(_n, ([tuple(t.shape) for t in inp], tuple(out.shape)))))
    with torch.no_grad():
        net(image)
    for h in hooks:
        h.remove()
    return shapes


def production_profile():
    """The newest offline profile of the production step committed under profiles/ (tools/prof_join.py: rocprofv3 kernel trace +
    separate FETCH_SIZE / WRITE_SIZE passes of `bench.py --profile`, joined launch by launch with the records of the engine).
    rocprofv3 --pmc cannot run inside this process, so `roofline.traffic` comes from there -- for the SAME launches (kernel
    family x stage of the pruned pass) the live numbers of this line describe."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_production_by_stage.json")))
    if not paths:
        return None
    try:
        d = json.load(open(paths[-1]))
    except Exception:
        return None
    d["source"] = os.path.relpath(paths[-1], ROOT)
    # a profile describes the code it was taken on: stale once a kernel or the host side changed (the hash of
    # p4v_kernels.h + p4v_api.hip + the header is stamped by tools/prof_join.py)
    from ptq4vit_amd import _lib
    d["current"] = d.get("source_hash") is not None and d.get("source_hash") == _lib.source_hash()
    return d


def aggregate_launches(recs, peak_i8=5000.0, peak_f32=157.3):
    """Per kernel family, and per stage of the pruned search pass inside it, from the engine's per-launch records (HIP events on
    the launch stream): launches, total ms, average launch, algorithmic ops per launch, achieved / issued TOP/s, fraction of the
    dense peak of the family's MFMA type (int8: 2 x the 2.5 PF bf16 dense spec; fp32: 157.3 TF -- MI355X_MICROARCH.md)."""
    fams = {}
    for r in recs:
        f = fams.setdefault(r["kernel"], {"launches": 0, "ms": 0.0, "ops": 0.0, "alg_ops": 0.0, "alg_bytes": 0.0, "by_stage": {}})
        st = f["by_stage"].setdefault(r["stage"], {"launches": 0, "ms": 0.0, "ops": 0.0, "alg_ops": 0.0, "alg_bytes": 0.0, "empty_launches": 0})
        for d in (f, st):
            d["launches"] += 1
            d["ms"] += r["ms"]
            d["ops"] += r["ops"]
            d["alg_ops"] += r["alg_ops"]
            d["alg_bytes"] += r.get("alg_bytes", 0.0)
        st["empty_launches"] += int(r["alg_ops"] == 0.0)

    def fin(d, peak):
        secs = d["ms"] * 1e-3
        d["avg_launch_ms"] = d["ms"] / d["launches"]
        d["ops_per_launch"] = d["alg_ops"] / d["launches"]
        d["algorithmic_bytes_per_launch"] = d.pop("alg_bytes") / d["launches"]
        d["achieved"] = d["alg_ops"] / secs / 1e12 if secs > 0 else None
        d["issued"] = d["ops"] / secs / 1e12 if secs > 0 else None
        d["frac"] = d["achieved"] / peak if secs > 0 else None
        d["issued_frac"] = d["issued"] / peak if secs > 0 else None
    for name, f in fams.items():
        peak = peak_f32 if ("float" in name or "sos" in name) else peak_i8
        f["peak"], f["unit"] = peak, "TOP/s" if peak == peak_i8 else "TFLOP/s"
        fin(f, peak)
        for st in f["by_stage"].values():
            fin(st, peak)
    return fams


def _rccl_version():
    try:
        return ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        return None


def _cpu_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        from threadpoolctl import threadpool_info
        threads = max([t.get("num_threads", 1) for t in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    return model, threads, os.cpu_count() or 1


# ViT-B/224 layer types: (count per network, builder of one seeded module at `imgs` images) -- SURVEY.md s8-d4 asks for
# per-layer-type timings x counts for the headline config (a full ViT-B run of the CPU path takes of the order of an hour)
def _cpu_layer_types(imgs, dim=768, heads=12, tokens=197, mlp=4, patch=16, img=224):
    from oracle.ptq4vit_oracle import ConvOracle, LinearOracle, MatMulOracle
    rng = np.random.default_rng(0)
    hp = dict(metric="hessian", eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=1)
    f32 = np.float32

    def linear(K, N, n_V, gelu=False, rows=None):
        w = (rng.standard_normal((N, K)) * 0.02).astype(f32)
        b = np.zeros(N, f32)
        x = rng.standard_normal((imgs, tokens, K) if rows is None else (imgs, K)).astype(f32)
        if gelu:
            x = (0.5 * x * (1.0 + np.tanh(0.7978845608 * (x + 0.044715 * x ** 3)))).astype(f32)
        out = (x @ w.T + b).astype(f32)
        grad = (rng.standard_normal(out.shape) * 1e-3).astype(f32)
        o = LinearOracle(w, b, w_bit=8, a_bit=8, n_V=n_V, postgelu=gelu, chunk=16, **hp)
        return (lambda: o.calibration_step2(x, out, grad)), 2.0 * 100 * x.size / K * K * N

    def matmul(sos):
        D = dim // heads
        if sos:
            A = rng.standard_normal((imgs, heads, tokens, tokens)).astype(f32) * 3
            A = np.exp(A - A.max(-1, keepdims=True))
            A = (A / A.sum(-1, keepdims=True)).astype(f32)
            B = rng.standard_normal((imgs, heads, tokens, D)).astype(f32)
        else:
            A = (rng.standard_normal((imgs, heads, tokens, D)) * D ** -0.5).astype(f32)
            B = rng.standard_normal((imgs, heads, D, tokens)).astype(f32)
        out = (A @ B).astype(f32)
        grad = (rng.standard_normal(out.shape) * 1e-3).astype(f32)
        o = MatMulOracle(A_bit=8, B_bit=8, sos=sos, chunk=4, **hp)
        per = imgs * heads * A.shape[2] * A.shape[3] * B.shape[3]
        return (lambda: o.calibration_step2(A, B, out, grad)), ((20 + 100) if sos else 200) * per

    def conv():
        w = (rng.standard_normal((dim, 3, patch, patch)) * 0.02).astype(f32)
        b = np.zeros(dim, f32)
        x = rng.standard_normal((imgs, 3, img, img)).astype(f32)
        o = ConvOracle(w, b, stride=patch, w_bit=8, a_bit=32, channelwise=True, **hp)
        from oracle.ptq4vit_oracle import im2col
        cols, fh, fw = im2col(x, (patch, patch), (patch, patch), (0, 0), (1, 1))
        out = (cols @ w.reshape(dim, -1).T).transpose(0, 2, 1).reshape(imgs, dim, fh, fw).astype(f32)
        grad = (rng.standard_normal(out.shape) * 1e-3).astype(f32)
        return (lambda: o.calibration_step2(x, out, grad)), 100.0 * imgs * fh * fw * 3 * patch * patch * dim

    depth = 12
    return [("qkv", depth, lambda: linear(dim, 3 * dim, 3)), ("proj", depth, lambda: linear(dim, dim, 1)),
            ("fc1", depth, lambda: linear(dim, mlp * dim, 1)), ("fc2 (post-GELU twin)", depth, lambda: linear(mlp * dim, dim, 1, gelu=True)),
            ("matmul1 q.k", depth, lambda: matmul(False)), ("matmul2 attn.v (split-of-softmax)", depth, lambda: matmul(True)),
            ("patch-embed conv (channel-wise)", 1, conv), ("head", 1, lambda: linear(dim, 1000, 1, rows=1))]


def _cpu_layer_types_torch(imgs, dim=768, heads=12, tokens=197, mlp=4, patch=16, img=224):
    """The same eight layer types through oracle/torch_port.py: torch operators on all host threads (what the reference's
    CPU path uses -- F.linear / @ / F.conv2d and multi-threaded elementwise kernels)."""
    from oracle.torch_port import TorchConv, TorchLinear, TorchMatMul
    g = torch.Generator().manual_seed(0)
    hp = dict(metric="hessian", eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=1)
    rn = lambda *s: torch.randn(*s, generator=g)

    def linear(K, N, n_V, gelu=False, rows=None):
        w, b = rn(N, K) * 0.02, torch.zeros(N)
        x = rn(imgs, tokens, K) if rows is None else rn(imgs, K)
        if gelu:
            x = torch.nn.functional.gelu(x)
        out = torch.nn.functional.linear(x, w, b)
        grad = rn(*out.shape) * 1e-3
        o = TorchLinear(w, b, w_bit=8, a_bit=8, n_V=n_V, postgelu=gelu, chunk=10 if N >= 2048 else 4, **hp)
        return (lambda: o.calibration_step2(x, out, grad)), 2.0 * 100 * x.numel() * N

    def matmul(sos):
        D = dim // heads
        if sos:
            A, B = torch.softmax(rn(imgs, heads, tokens, tokens) * 3, -1), rn(imgs, heads, tokens, D)
        else:
            A, B = rn(imgs, heads, tokens, D) * D ** -0.5, rn(imgs, heads, D, tokens)
        out = A @ B
        grad = rn(*out.shape) * 1e-3
        o = TorchMatMul(A_bit=8, B_bit=8, sos=sos, chunk=2, **hp)
        per = imgs * heads * A.shape[2] * A.shape[3] * B.shape[3]
        return (lambda: o.calibration_step2(A, B, out, grad)), ((20 + 100) if sos else 200) * per

    def conv():
        w, b, x = rn(dim, 3, patch, patch) * 0.02, torch.zeros(dim), rn(imgs, 3, img, img)
        out = torch.nn.functional.conv2d(x, w, b, patch)
        grad = rn(*out.shape) * 1e-3
        o = TorchConv(w, b, stride=patch, w_bit=8, a_bit=32, channelwise=True, **hp)
        return (lambda: o.calibration_step2(x, out, grad)), 100.0 * imgs * out.shape[2] * out.shape[3] * 3 * patch * patch * dim

    depth = 12
    return [("qkv", depth, lambda: linear(dim, 3 * dim, 3)), ("proj", depth, lambda: linear(dim, dim, 1)),
            ("fc1", depth, lambda: linear(dim, mlp * dim, 1)), ("fc2 (post-GELU twin)", depth, lambda: linear(mlp * dim, dim, 1, gelu=True)),
            ("matmul1 q.k", depth, lambda: matmul(False)), ("matmul2 attn.v (split-of-softmax)", depth, lambda: matmul(True)),
            ("patch-embed conv (channel-wise)", 1, conv), ("head", 1, lambda: linear(dim, 1000, 1, rows=1))]


def cpu_baseline(calib=32, rounds=3, sample_images=4, backend="torch"):
    """CPU path beside the GPU number: a port of the reference algorithm pinned to the reference by
    tests/golden -- `backend` "torch": oracle/torch_port.py, torch operators on all host threads, the way the reference's own
    CPU path runs; "numpy": the parity oracle (single-threaded elementwise passes) -- timed on this box's host cores, one
    search round of EVERY ViT-B/224 layer type at
    `sample_images` images, scaled to the headline workload: x (calib / sample_images) images (the work is linear in
    the rows), x `rounds`, x layer counts.  Search only -- the reference's CPU path would add 74 x 8 capture passes.
    Returns (estimated seconds for one ViT-B/224 calibration, per-type table, seconds spent sampling)."""
    table = []
    total = spent = 0.0
    for name, count, build in (_cpu_layer_types_torch if backend == "torch" else _cpu_layer_types)(sample_images):
        run, macs = build()
        t = time.time()
        run()
        dt = time.time() - t
        spent += dt
        est = dt * (calib / sample_images) * rounds * count
        total += est
        table.append({"layer": name, "count": count, "sample_s": round(dt, 3), "sample_tmacs": round(macs / 1e12, 4),
                      "cpu_tflops": round(2.0 * macs / dt / 1e12, 3), "est_full_s": round(est, 1)})
    return total, table, spent


def cpu_full_deit_tiny(calib=4):
    """BASELINE.json config 0 on the CPU path, in full: DeiT-tiny/224, BasePTQ (cosine, 1 round), 4 calibration images --
    every wrapped module's step 2 through the numpy oracle on tensors captured by a CPU forward/backward of the same
    network (capture time reported separately).  ~1 min of host time: run with `bench.py --cpu-full`, not by default."""
    import contextlib
    import io
    from oracle.ptq4vit_oracle import ConvOracle, LinearOracle, MatMulOracle
    from ptq4vit_amd.configs import BasePTQ
    from ptq4vit_amd.quant_layers.conv import MinMaxQuantConv2d
    from ptq4vit_amd.quant_layers.linear import MinMaxQuantLinear
    from ptq4vit_amd.utils import models, net_wrap
    from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator
    net = models.get_net("deit_tiny_patch16_224", seed=0, device="cpu")
    with contextlib.redirect_stdout(io.StringIO()):
        wrapped = net_wrap.wrap_modules_in_net(net, BasePTQ)
    images = torch.randn(calib, 3, 224, 224, generator=torch.Generator().manual_seed(0))
    t_search = [0.0]
    npy = lambda t: None if t is None else t.detach().numpy()
    for m in wrapped.values():
        def step2(_m=m):
            hp = dict(metric=_m.metric, eq_alpha=_m.eq_alpha, eq_beta=_m.eq_beta, eq_n=_m.eq_n, search_round=_m.search_round)
            t = time.time()
            if isinstance(_m, MinMaxQuantLinear):
                LinearOracle(npy(_m.weight), npy(_m.bias), w_bit=_m.w_bit, a_bit=_m.a_bit, n_V=_m.n_V,
                             postgelu=type(_m).__name__.startswith("PostGelu"), **hp).calibration_step2(
                    npy(_m.raw_input), npy(_m.raw_out), npy(_m.raw_grad))
            elif isinstance(_m, MinMaxQuantConv2d):
                ConvOracle(npy(_m.weight), npy(_m.bias), stride=_m.stride, w_bit=_m.w_bit, a_bit=32,
                           channelwise=type(_m).__name__.startswith("Channelwise"), **hp).calibration_step2(
                    npy(_m.raw_input), npy(_m.raw_out), npy(_m.raw_grad))
            else:
                MatMulOracle(A_bit=_m.A_bit, B_bit=_m.B_bit, sos=type(_m).__name__.startswith("SoS"), **hp).calibration_step2(
                    npy(_m.raw_input[0]), npy(_m.raw_input[1]), npy(_m.raw_out), npy(_m.raw_grad))
            t_search[0] += time.time() - t
            _m.calibrated = True
            del _m.raw_input, _m.raw_out, _m.raw_grad
        m.calibration_step2 = step2
    cal = HessianQuantCalibrator(net, wrapped, SyntheticLoader(images), sequential=False, batch_size=4)
    t = time.time()
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        cal.batching_quant_calib()
    wall = time.time() - t
    return {"config": "DeiT-tiny/224 BasePTQ W8A8, 4 calibration images, 74 modules (BASELINE.json config 0)",
            "wall_s": round(wall, 2), "search_s": round(t_search[0], 2), "capture_s": round(wall - t_search[0], 2),
            "layers_per_s": round(len(wrapped) / wall, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="vit_base_patch16_224")
    ap.add_argument("--calib", type=int, default=32)
    ap.add_argument("--config", default="PTQ4ViT", choices=["PTQ4ViT", "BasePTQ"],
                    help="quantisation config module (reference configs/): PTQ4ViT = twin quantisers + hessian metric, 3 rounds (the headline); "
                         "BasePTQ = cosine metric, 1 round, plain quantisers")
    ap.add_argument("--bits", type=int, default=8, help="W/A bit width of every wrapped module (8 = headline W8A8; 6 = the W6A6 config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-numpy", action="store_true", help="time the numpy parity oracle as the CPU baseline instead of the multi-threaded torch port")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extras (fresh-network calibration, quant_forward throughput)")
    ap.add_argument("--cpu-full", action="store_true", help="also run BASELINE config 0 (DeiT-tiny/224 BasePTQ x 4 images) through the CPU oracle, in full")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--search-streams", type=int, default=0, help="host threads / HIP streams of the module searches in the timed steps "
                    "(0 = the calibrator's default of 4; 1 = serial, the launch order of the roofline step)")
    ap.add_argument("--profile", action="store_true",
                    help="for runs under rocprofv3: every calibration of the process is the SAME single-stream production step (no "
                         "single-pass / fresh-network / full-sweep / CPU extras), so that a kernel trace divides into identical calibrations")
    ap.add_argument("--dump-launches", default="", help="write the per-launch records of the roofline step (JSON) for tools/prof_join.py")
    ap.add_argument("--capture-batch", type=int, default=0,
                    help="images per capture pass of the timed steps (0 = batch_size = 4, the reference's passes)")
    ap.add_argument("--tune", default="", help="key=value,... engine tuning overrides (p4v_debug_set_tuning), experiments only")
    ap.add_argument("--variant", type=int, default=0, help="engine A/B switch word (p4v_debug_set_variant), experiments only: use with --no-roofline")
    args = ap.parse_args()

    if args.profile:
        args.search_streams, args.no_extras, args.no_cpu_baseline = 1, True, True
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("P4V_FORCE_DEVICE", os.environ.get("LOCAL_RANK", 0)))   # P4V_FORCE_DEVICE: testing N ranks on one GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("P4V_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm; gloo only for single-GPU testing
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from ptq4vit_amd import engine
    import importlib
    PTQ4ViT = importlib.import_module(f"ptq4vit_amd.configs.{args.config}")      # (the name the rest of this file uses for "the config")
    from ptq4vit_amd.utils import models, net_wrap
    from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator

    for kv in filter(None, args.tune.split(",")):
        engine.debug_tuning(*(int(v) for v in kv.split("=")))
    if args.variant:
        engine.debug_variant(args.variant)
    net = models.get_net(args.model, seed=0, device=dev)
    if args.bits != 8:      # what the reference's drivers do to the config module (example/test_all.py:53-78)
        PTQ4ViT.bit = args.bits
        for tab in (PTQ4ViT.w_bit, PTQ4ViT.a_bit, PTQ4ViT.A_bit, PTQ4ViT.B_bit):
            for k in tab:
                tab[k] = args.bits
    import contextlib as _ctx, io as _io
    with _ctx.redirect_stdout(_io.StringIO()):       # (stdout carries the ONE JSON line and nothing else)
        wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
    img = models.input_size(args.model)
    g = torch.Generator(device="cpu").manual_seed(0)
    images = torch.randn(args.calib, 3, img, img, generator=g).to(dev)
    loader = SyntheticLoader(images)
    shapes = probe_shapes(net, wrapped, images[:1])

    def calibrate(net_, wrapped_, search_streams=None, capture_batch=None):
        for m in wrapped_.values():
            m.mode = "raw"
        cal = HessianQuantCalibrator(net_, wrapped_, loader, sequential=False, batch_size=4,
                                     capture_batch_size=capture_batch or args.capture_batch or None)
        if search_streams or args.search_streams:
            cal.search_streams = search_streams or args.search_streams
        cal.batching_quant_calib()
        return cal

    def one_step(search_streams=None, capture_batch=None):      # re-calibration of the network built above (roofline / extras)
        return calibrate(net, wrapped, search_streams, capture_batch)

    def fresh_pair():
        net_ = models.get_net(args.model, seed=0, device=dev)
        return net_, net_wrap.wrap_modules_in_net(net_, PTQ4ViT)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    import io
    import contextlib
    quiet = contextlib.redirect_stdout(io.StringIO())
    # A STEP is what the reference's driver times (example/test_all.py:24-34): a NEW network object -- new parameter storage,
    # freshly wrapped -- calibrated ONCE by HessianQuantCalibrator(...).batching_quant_calib().  The networks are built before the
    # clock starts (the reference builds and wraps outside its timed region too); `--profile` runs re-calibrate one network, so
    # that a kernel trace divides into identical calibrations.
    with quiet:
        pairs = [(net, wrapped)] * (args.warmup + args.steps) if args.profile else [fresh_pair() for _ in range(args.warmup + args.steps)]
        sync()
        t_cold = time.time()
        cold = None
        for i in range(args.warmup):
            calibrate(*pairs[i])
            if i == 0:
                sync()
                cold = time.time() - t_cold   # first calibration of this process: eager capture, cold kernels
        sync()
        engine.launch_counters(reset=True)
        t0 = time.time()
        cals = [calibrate(*pairs[args.warmup + i]) for i in range(args.steps)]
        sync()
        elapsed = time.time() - t0
        launches = engine.launch_counters(reset=True)
        del pairs
        # the same network object calibrated again and again (rounds 1-5 timed this): its capture graph needs no parameter copy
        steady_s = None
        if not args.profile:
            one_step(); one_step()
            sync()
            t_s = time.time()
            for _ in range(3):
                one_step()
            sync()
            steady_s = (time.time() - t_s) / 3
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # the same calibration with ONE capture pass over all images (opt-in `capture_batch_size`: not part of `value`)
    cal_big, big_s = None, None
    if not args.profile:
        with quiet:
            one_step(capture_batch=args.calib)
            sync()
            t_big = time.time()
            cal_big = one_step(capture_batch=args.calib)
            sync()
            big_s = time.time() - t_big
    n_mod = len(wrapped)
    value = n_mod * args.steps / elapsed

    roof = None
    if not args.no_roofline:
        # live HIP-event timing of the dominant kernels (the int8 sweeps) on their launch stream, over one more
        # step that is not part of `value` (every rank runs it: the step ends with a collective)
        engine.stats_reset()
        engine.prune_counters(reset=True)
        engine.stats_enable(rank == 0)
        with quiet:
            cal_r = one_step(search_streams=1)   # kernels timed in isolation: no second stream sharing the CUs
        sync()
        st = engine.stats_get()
        recs = engine.stats_launches() if rank == 0 else []
        engine.stats_enable(False)
        if rank == 0 and st["sweep_i8_launches"] > 0:
            mine = {n: m for n, m in wrapped.items() if cal_r.owner[n] == rank}
            lin, mm, conv = search_macs(mine, shapes, args.calib)
            peak = 5000.0  # TOP/s: dense int8 MFMA = 2x the 2.5 PF bf16 dense spec (MI355X_MICROARCH.md)
            fams = aggregate_launches(recs, peak)
            if args.dump_launches:
                with open(args.dump_launches, "w") as fh:
                    json.dump({"model": args.model, "bits": args.bits, "calib": args.calib, "launches": recs}, fh)
            i8 = {k: v for k, v in fams.items() if v["peak"] == peak}
            dom_name = max(i8, key=lambda k: i8[k]["ms"])
            dom = i8[dom_name]
            algo_ops = 2.0 * st["sweep_i8_alg_macs"]   # ops of the reference GEMMs the EXECUTED int8 launches stand for (unpadded,
            ref_ops = 2.0 * (lin + mm)                 # one plane per candidate); memo-restored passes are not launched, not counted
            issued_ops = 2.0 * st["sweep_i8_macs"]     # incl. tile padding and the second twin plane
            secs = st["sweep_i8_ms"] * 1e-3
            all_alg = sum(f["alg_ops"] for f in fams.values())      # int8 + fp32 sweeps (patch embedding, split search)
            search_s = sum(c.timings["search_s"] for c in cals) / len(cals)
            prof = production_profile()
            pk = (prof or {}).get("kernels", {}).get(dom_name)
            fresh = bool(prof and prof.get("current"))
            roof = {"bound": "mfma", "kernel": dom_name + " (int8 candidate sweep; the kernel family with the most time in a calibration)",
                    "achieved": dom["achieved"], "peak": peak, "unit": "TOP/s", "frac": dom["frac"],
                    # `peak` = 2 x the 2.5 PF bf16 dense spec (the guide lists no int8 spec); the guide's only measured int8
                    # figure is the 16x16x64 micro-benchmark ceiling of 3944 TOP/s: the fraction of that, beside it
                    "peak_guide_ubench": 3944.0, "frac_of_guide_ubench": dom["achieved"] / 3944.0,
                    # HBM-side bytes per launch of THIS kernel family over THE SAME launches (the production step, one stream),
                    # from the separate rocprofv3 --pmc passes joined by tools/prof_join.py: FETCH_SIZE x 2 (gfx950 wide-read
                    # correction) + WRITE_SIZE; null until a profile of this code is committed
                    "traffic": pk.get("traffic_bytes_per_launch") if (pk and fresh) else None,
                    "traffic_note": None if (pk and fresh) else ("no PMC profile of this code under profiles/ (newest: %s, taken on other sources)" % prof["source"] if prof else "no profile committed"),
                    "launches": dom["launches"], "avg_launch_ms": dom["avg_launch_ms"], "ops_per_launch": dom["ops_per_launch"],
                    "issued": dom["issued"],
                    # the same instruction alone (MFMA-only loop, 8 waves/CU, tools/ubench_mfma.hip) sustains 4044 TOP/s on
                    # this part at its ~2.0 GHz clock under load (profiles/r1_ubench.txt); `peak` stays the 2 x bf16 spec
                    "peak_measured_mfma_only": 4044.0, "frac_of_measured": dom["achieved"] / 4044.0,
                    "event_overhead_ms": st["event_overhead_ms"],   # an empty event pair, subtracted from every launch duration
                    "measured": "HIP events around every sweep launch of one untimed single-stream calibration (same launches as the "
                                "timed steps: exact pruning + memo on); per-launch records -> by_kernel / by_stage",
                    # every sweep family of the step; by_stage: A = all candidates on the sample slice, B1 = the bound (one
                    # candidate or the hull of the slice winners, all samples), B2 = the survivors, full = unpruned passes
                    "by_kernel": fams,
                    "all_int8_sweeps": {"achieved": algo_ops / secs / 1e12, "issued": issued_ops / secs / 1e12,
                                        "frac": algo_ops / secs / 1e12 / peak, "launches": st["sweep_i8_launches"],
                                        "avg_launch_ms": st["sweep_i8_ms"] / st["sweep_i8_launches"]},
                    # the whole search phase of the TIMED steps (4 streams, every kernel and gap included): executed algorithmic
                    # ops of all sweeps / search_s
                    "whole_search": {"executed_ops": all_alg, "search_s": search_s, "achieved": all_alg / search_s / 1e12,
                                     "frac": all_alg / search_s / 1e12 / peak},
                    "reference_ops_fraction_executed": algo_ops / ref_ops,
                    "memo_hits": st["memo_hits"], "memo_misses": st["memo_misses"],
                    "prune_counters": engine.prune_counters(),
                    # offline cross-check: the same kernel family in the committed rocprofv3 profile of `bench.py --profile`
                    "profile": None if not pk else {"source": prof["source"], "taken_on_this_code": fresh, "launches_per_calibration": pk.get("launches_per_calibration"),
                                                     "avg_launch_ms": pk.get("avg_launch_ms"), "frac": pk.get("frac"),
                                                     "traffic_bytes_per_launch": pk.get("traffic_bytes_per_launch"),
                                                     "algorithmic_bytes_per_launch": pk.get("algorithmic_bytes_per_launch"),
                                                     # SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / launch cycles at 2.4 GHz (PMC pass)
                                                     "mfma_busy_of_spec_cycles": pk.get("mfma_busy_of_spec_cycles")}}

    # ---- untimed extras (not part of `value`) ---------------------------------------------------------------------------
    eager_s = qf = None
    if world == 1 and not args.no_extras:
        # (a) a fresh network with the architecture's capture graph switched off (P4V_ARCH_GRAPHS=0): eight eager sub-batch passes,
        # what every fresh network cost until round 5 and what the first one of a process still costs
        with quiet:
            net2, wrapped2 = fresh_pair()
            sync()
            os.environ["P4V_ARCH_GRAPHS"] = "0"
            t_f = time.time()
            try:
                calibrate(net2, wrapped2)
                sync()
            finally:
                os.environ.pop("P4V_ARCH_GRAPHS", None)
            eager_s = time.time() - t_f
        del net2, wrapped2
        # (b) the calibrated network as an inference path (reference example/test_vit.py:26-45 evaluates 50 k images through
        # quant_forward): images / s at batch 128 on the int8 path, next to the raw fp32 forward of the same network
        try:
            xb = torch.randn(128, 3, img, img, generator=torch.Generator().manual_seed(1)).to(dev)

            def rate(reps=3):
                with torch.no_grad():
                    net(xb[:8]); net(xb)
                    sync()
                    t_ = time.time()
                    for _ in range(reps):
                        net(xb)
                    sync()
                return reps * xb.shape[0] / (time.time() - t_)
            q_rate = rate()
            for m in wrapped.values():
                m.mode = "raw"
            r_rate = rate()
            for m in wrapped.values():
                m.mode = "quant_forward"
            qf = {"quant_forward_img_s": q_rate, "raw_fp32_forward_img_s": r_rate, "batch": 128, "ratio": q_rate / r_rate}
            del xb
        except torch.cuda.OutOfMemoryError:
            qf = None

    # N > 1: both capture plans measured side by side (three fresh networks each, the fastest; max over ranks), the all_to_all
    # rate the automatic choice was made on, and what tools/predict_scale.py predicted for this world size from one-GPU
    # measurements -- a first SCALE run needs no re-run to be interpreted
    capture_modes = predicted = None
    if world > 1 and not args.profile:
        from ptq4vit_amd.utils import shard as _shard
        capture_modes = {"automatic_choice": getattr(cals[-1], "capture_mode", None), "a2a_gbps_per_peer_measured": _shard.a2a_rate_gbps()}
        for mode_name, flag in (("replicated", False), ("sharded", True)):
            best_t, tm_ = None, None
            for _ in range(3):
                with quiet:
                    n_, w_ = fresh_pair()
                    cal_m = HessianQuantCalibrator(n_, w_, loader, sequential=False, batch_size=4)
                    cal_m.shard_capture = flag
                    sync()
                    t_m = time.time()
                    cal_m.batching_quant_calib()
                    sync()
                    dt_m = time.time() - t_m
                tt_m = torch.tensor([dt_m], device=dev, dtype=torch.float64)
                dist.all_reduce(tt_m, op=dist.ReduceOp.MAX)
                if best_t is None or float(tt_m.item()) < best_t:
                    best_t, tm_ = float(tt_m.item()), dict(cal_m.timings, capture_mode=cal_m.capture_mode)
                del n_, w_, cal_m
            capture_modes[mode_name] = {"step_s": best_t, "layers_per_s": n_mod / best_t, "mode_that_ran": tm_["capture_mode"],
                                        "capture_s": tm_["capture_s"], "search_s": tm_["search_s"], "exchange_s": tm_.get("exchange_s")}
        import glob
        preds = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_scale_prediction.json")))
        if preds:
            try:
                pd_ = json.load(open(preds[-1]))
                for cfg_ in pd_.values():
                    if cfg_["measured"]["model"] == args.model and cfg_["measured"]["calib_images"] == args.calib and cfg_["measured"]["bits"] == args.bits:
                        row_ = next((r for r in cfg_["prediction"]["rows"] if r["world"] == world), None)
                        if row_:
                            predicted = {"source": os.path.relpath(preds[-1], ROOT), "layers_per_s": row_["layers_per_s"], "step_s": row_["step_s"],
                                         "capture_mode": row_["capture_mode"]}
            except Exception:      # noqa: BLE001 - a stale or foreign file must not break the bench line
                predicted = None

    per_rank = None
    if world > 1:
        # first SCALE run diagnosable: every rank's share of the last timed step
        tm = cals[-1].timings
        mine_t = {"rank": rank, "owned": tm.get("owned"), "capture_s": tm["capture_s"], "search_s": tm["search_s"],
                  "exchange_s": tm.get("exchange_s", 0.0), "total_s": tm["total_s"]}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine_t)

    # the same kernels on FULL sweeps (every candidate over every sample: exact pruning and nothing else switched off, one
    # more untimed single-stream step): what the sweep kernels sustain when a launch is not a short candidate range
    if roof is not None and world == 1 and not args.profile:
        engine.debug_variant(4194304)
        engine.stats_reset()
        engine.stats_enable(True)
        with quiet:
            one_step(search_streams=1)
        sync()
        engine.stats_get()
        full = aggregate_launches(engine.stats_launches())
        engine.stats_enable(False)
        engine.debug_variant(0)
        keep = ("launches", "ms", "avg_launch_ms", "ops_per_launch", "achieved", "issued", "frac", "issued_frac")
        roof["full_sweeps_without_pruning"] = {k: {q: v[q] for q in keep} for k, v in full.items()}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_model, threads, logical = _cpu_info()
        backend = "numpy" if args.cpu_numpy else "torch"
        if backend == "torch":
            # thread count from tools/cpu_port_threads.py on the GPU box (2 x EPYC 9575F, 256 logical CPUs): at the sample's
            # sizes (788 rows) 16-32 threads are the optimum; torch's default of 128 is 3-10 x slower (thread start-up per op)
            threads = min(32, os.cpu_count() or 1)
            torch.set_num_threads(threads)
        est_s, per_type, spent = cpu_baseline(calib=args.calib, backend=backend)
        what = ("oracle/torch_port.py (torch-CPU restatement of the reference's calibration_step2 on all host threads, validated "
                "against the reference's golden files)" if backend == "torch" else
                "numpy oracle (port of the reference's calibration_step2, pinned by tests/golden)")
        cpu = {"value": n_mod / est_s, "unit": "layers/s", "cores": threads, "kind": "port", "port_backend": backend,
               "cpu_model": cpu_model, "logical_cpus": logical, "est_calibration_s": round(est_s, 1),
               "sample": f"{what}, ONE search round of each "
                         f"ViT-B/224 layer type at 4 images ({spent:.1f} s of CPU work), scaled x{args.calib / 4:g} images x 3 rounds x layer "
                         f"counts to the 74-module workload; search only (the reference's CPU path adds 74 x 8 capture passes)",
               "per_layer_type": per_type}
        if args.cpu_full:
            cpu["config0_full_run"] = cpu_full_deit_tiny()

    # post-quant ImageNet top-1 (BASELINE.json north_star; reference example/test_vit.py:26-45): measured when the machine has
    # the data -- P4V_IMAGENET = ImageNet root (train/ + val/), P4V_WEIGHTS = timm checkpoint of --model -- through
    # tools/eval_top1.py (FP32 and quantised top-1 of the pretrained network, calibrated on 32 train images drawn with the
    # reference's seed-3 rule).  Untimed, after everything else.
    top1, top1_reason = None, ("no ImageNet and no pretrained weights in this environment (no network): set P4V_IMAGENET and P4V_WEIGHTS "
                               "to fill this through tools/eval_top1.py; parity evidence without them is interval parity on identical "
                               "tensors + the reference's own quantised logits (tests/golden: mini ViT, DeiT-tiny/224)")
    if rank == 0 and world == 1 and os.environ.get("P4V_IMAGENET") and os.environ.get("P4V_WEIGHTS"):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import eval_top1
        del net, wrapped
        torch.cuda.empty_cache()
        max_val = int(os.environ["P4V_TOP1_MAX_VAL"]) if os.environ.get("P4V_TOP1_MAX_VAL") else None
        top1 = eval_top1.evaluate(args.model, os.environ["P4V_IMAGENET"], os.environ["P4V_WEIGHTS"], args.config, args.bits, args.calib,
                                  3, 128, max_val, int(os.environ.get("P4V_TOP1_WORKERS", "8")), quiet=True)
        top1_reason = None

    if rank == 0:
        t = cals[-1].timings
        line = {
            "metric": "calibration throughput (wrapped modules calibrated per second), ViT-B/224 W8A8, 32 calibration images" if (args.model == "vit_base_patch16_224" and args.bits == 8 and args.calib == 32 and args.config == "PTQ4ViT") else f"calibration throughput (wrapped modules calibrated per second), {args.model} {args.config} W{args.bits}A{args.bits}, {args.calib} calibration images",
            "value": value, "unit": "layers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "calibration_wall_clock_s": elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int8",
            "data": "synthetic (seeded N(0,1) images, trunc-normal weights)",
            "config": {"workload": f"{args.model} {args.config} W{args.bits}A{args.bits}, {args.calib} calibration images, {n_mod} wrapped modules; every step "
                                   "calibrates a FRESH network object once: HessianQuantCalibrator(net, wrapped, loader, sequential=False, "
                                   "batch_size=4).batching_quant_calib() (capture + search), the region reference example/test_all.py:31-34 times",
                       "global_batch": args.calib, "parallelism": f"layers sharded over {world} GPU(s)"},
            "breakdown": {"capture_s": t["capture_s"], "search_s": t["search_s"]},
            # images per capture pass of the timed steps (batch_size = 4: the reference's passes), and the same calibration
            # with one pass over all images (opt-in, utils/quant_calib.py: see the caveat on raw_grad there)
            "capture": {"images_per_pass": cals[-1]._capture_bs(),
                        "opt_in_single_pass": None if cal_big is None else {
                            "images_per_pass": cal_big._capture_bs(), "calibration_wall_clock_s": big_s,
                            "capture_s": cal_big.timings["capture_s"], "search_s": cal_big.timings["search_s"]}},
            # every timed step calibrates a NEW network object (new parameter storage, fresh wrap): the sub-batch passes replay the
            # HIP graph recorded for the ARCHITECTURE (second network of an architecture in the process on: utils/quant_calib.py,
            # _arch_shadow) after copying the new network's parameters into the graph's storage.  The first calibration of the
            # process (cold kernels, eager capture), untimed:
            "first_calibration_s": cold,
            "fresh_network_calibration_s": elapsed / args.steps,          # = calibration_wall_clock_s (kept under its round-5 name)
            # the SAME network object calibrated again and again (what rounds 1-5 reported as `value`), and a fresh network without
            # the architecture's graph (eager sub-batch passes: rounds 1-5's fresh-network figure)
            "steady_state_recalibration_s": steady_s,
            "fresh_network_eager_capture_s": eager_s,
            # kernel launches of the library per calibration in the timed steps: asked for by the per-module code / issued to the
            # GPU after grouping (p4v_calibrate_group) / stream synchronisations of the groups
            "launches_per_calibration": {k: v / args.steps for k, v in launches.items()},
            # post-quant ImageNet top-1 (BASELINE.json north_star, reference example/test_vit.py:26-45), see above
            "top1": top1, "top1_reason": top1_reason,
            "quant_forward_img_s": qf["quant_forward_img_s"] if qf else None, "quant_forward": qf,
            "per_rank": per_rank,
            # a first SCALE run is self-diagnosing: what the ranks talked through, which capture mode the calibrator chose
            "distributed": {"world_size": world, "backend": (dist.get_backend() if world > 1 else None),
                            "rccl_version": _rccl_version(), "capture_mode": getattr(cals[-1], "capture_mode", None),
                            "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                            "capture_modes_measured": capture_modes, "predicted_from_one_gpu": predicted},
            "imbalance": (max(r["search_s"] for r in per_rank) / (sum(r["search_s"] for r in per_rank) / len(per_rank))) if per_rank else None,
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
