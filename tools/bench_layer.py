#!/usr/bin/env python
"""Micro-benchmark of one module's calibration on synthetic ViT-B shaped tensors (for rocprofv3 / tuning).

    python tools/bench_layer.py --layer fc1 --rounds 1 --reps 3
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ptq4vit_amd import engine  # noqa: E402

SHAPES = {  # name: (K, N, n_V, postgelu)
    "qkv": (768, 2304, 3, False), "proj": (768, 768, 1, False), "fc1": (768, 3072, 1, False),
    "fc2": (3072, 768, 1, True), "head": (768, 1000, 1, False),
    # ViT-L / Swin-B stage 4 (dim 1024) and ViT-S fc2: the other large-K layers (k_sweep7)
    "l_qkv": (1024, 3072, 3, False), "l_proj": (1024, 1024, 1, False), "l_fc1": (1024, 4096, 1, False),
    "l_fc2": (4096, 1024, 1, True), "s_fc2": (1536, 384, 1, True),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", default="fc1", help="one layer or a comma-separated list")
    ap.add_argument("--rounds", type=int, default=1)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=197)
    ap.add_argument("--bits", type=int, default=8)
    ap.add_argument("--variant", type=int, default=0, help="p4v_debug_set_variant word (kernel A/B switches)")
    ap.add_argument("--tune", default="", help="key=value,... launch-heuristic overrides (p4v_debug_set_tuning)")
    ap.add_argument("--metric", default="hessian", help="similarity metric (hessian, cosine, L2_norm, ...)")
    ap.add_argument("--kernel-stats", action="store_true", help="per-launch time of the sweep kernels (HIP events)")
    ap.add_argument("--vit-grad", action="store_true", help="raw_grad with the profile of a ViT under the reference's KL loss: magnitude 1e-10, "
                    "class-token rows > 99 %% of grad^2 (tests/test_hip_production_path.py::vit_like_grad) -- the pruned passes engage as in production")
    a = ap.parse_args()
    if "," in a.layer:
        for name in a.layer.split(","):
            a.layer = name
            one(a)
    else:
        one(a)


def one(a):
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    hp = dict(metric=a.metric, eq_alpha=0.01, eq_beta=1.2, eq_n=100, search_round=a.rounds)
    engine.debug_variant(a.variant)
    for kv in filter(None, a.tune.split(",")):
        k, v = kv.split("=")
        engine.debug_tuning(int(k), int(v))
    engine.stats_enable(a.kernel_stats)
    if a.layer in SHAPES:
        K, N, nV, gelu = SHAPES[a.layer]
        x = torch.randn(a.batch, a.tokens, K, generator=g)
        if gelu:
            x = torch.nn.functional.gelu(1.5 * x)
        w = torch.randn(N, K, generator=g) * 0.02
        b = torch.randn(N, generator=g) * 0.02
        x, w, b = x.to(dev), w.to(dev), b.to(dev)
        out = torch.nn.functional.linear(x, w, b)
        grad = (torch.randn(out.shape, generator=g) * 1e-3)
        if a.vit_grad:
            grad = torch.randn(out.shape, generator=g) * 1e-10
            grad[:, 0] *= 300.0 * torch.exp(torch.randn(out.shape[0], generator=g)).view(-1, 1)
        grad = grad.to(dev)
        run = lambda: engine.linear_calibrate(weight=w, bias=b, x=x, out=out, grad=grad, w_bit=a.bits, a_bit=a.bits,
                                              n_V=nV, n_H=1, n_a=1, postgelu=gelu, **hp)
        macs = 2.0 * 100 * a.rounds * a.batch * a.tokens * K * N
    elif a.layer in ("qk", "sv"):
        H, D, S = 12, 64, a.tokens
        if a.layer == "qk":
            A = torch.randn(a.batch, H, S, D, generator=g).to(dev)
            B = torch.randn(a.batch, H, S, D, generator=g).to(dev).transpose(-2, -1)
        else:
            A = torch.softmax(torch.randn(a.batch, H, S, S, generator=g) * 2, -1).to(dev)
            B = torch.randn(a.batch, H, S, D, generator=g).to(dev)
        out = A @ B
        grad = (torch.randn(out.shape, generator=g) * 1e-3).to(dev)
        run = lambda: engine.matmul_calibrate(A=A, B=B, out=out, grad=grad, A_bit=a.bits, B_bit=a.bits,
                                              sos=(a.layer == "sv"), **hp)
        macs = (2.0 if a.layer == "qk" else 1.2) * 100 * a.rounds * a.batch * H * S * S * D
    else:
        raise SystemExit("unknown layer")
    run()
    torch.cuda.synchronize()
    if a.kernel_stats:
        engine.stats_reset()
    t = time.time()
    for _ in range(a.reps):
        run()
    torch.cuda.synchronize()
    dt = (time.time() - t) / a.reps
    if a.kernel_stats:
        st = engine.stats_get()
        if st["sweep7_launches"]:
            nt, mt = st["sweep7_twin_launches"], st["sweep7_twin_ms"]
            n1, m1 = st["sweep7_launches"] - nt, st["sweep7_ms"] - mt
            print(f"{a.layer}: sweep7 plain: {n1} launches, {m1 / max(n1, 1) * 1e3:.1f} us each; twin: {nt} launches, {mt / max(nt, 1) * 1e3:.1f} us each")
        for k in ("sweep6", "sweep7", "sweep_i8", "sweep_f32"):
            n = st[k + "_launches"]
            if n:
                print(f"{a.layer}: {k}: {n} launches, {st[k + '_ms'] / n * 1e3:.1f} us each, "
                      f"{2 * st[k + '_alg_macs'] / (st[k + '_ms'] * 1e-3) / 1e12:.0f} TOP/s algorithmic")
    res = run()
    digest = "/".join(r.double().sum().item().hex() for r in res[:2] if r is not None)   # A/B builds must agree bit for bit
    print(f"{a.layer}: intervals {digest}")
    print(f"{a.layer}: {dt * 1e3:.2f} ms per calibration ({a.rounds} round(s)); {2 * macs / dt / 1e12:.1f} TOP/s algorithmic incl. pack/finish")


if __name__ == "__main__":
    main()
