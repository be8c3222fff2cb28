"""Glue between torch tensors and the C ABI: pointers, stream, workspace, candidate tables.

Nothing here computes: every calibration number comes out of libptq4vit_hip.so.
"""
import ctypes as C

import torch

from . import _lib

_workspace = {}
_mult_cache = {}


def _require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"ptq4vit_amd: {what} must live on the GPU (got {t.device}); the calibration path has no CPU fallback")


def device_of(*tensors):
    for t in tensors:
        if t is not None and t.is_cuda:
            return t.device
    if not torch.cuda.is_available():
        raise RuntimeError("ptq4vit_amd: no GPU visible -- the HIP calibration engine needs an MI355X (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def to_dev(t, dev):
    """fp32 tensor on `dev` (the reference keeps caches on the host and re-uploads per search call)."""
    if t is None:
        return None
    return t.detach().to(device=dev, dtype=torch.float32)


def workspace(dev, nbytes):
    """Scratch for one calibration call; one buffer per (device, stream) so that searches running on different
    streams (utils/quant_calib.py runs two modules at a time) never share scratch."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _workspace.get(key)
    if ws is None or ws.numel() < nbytes:
        _workspace.pop(key, None)
        ws = None
        ws = torch.empty(int(nbytes * 1.05) + (1 << 20), dtype=torch.uint8, device=dev)
        _workspace[key] = ws
    return ws


def release_workspace():
    _workspace.clear()


_side_streams = {}


def side_streams(dev, n):
    """`n` persistent side streams of `dev` (the calibrator searches several modules at a time).  Persistent because
    the scratch buffers are keyed by stream: fresh streams per calibration would strand one scratch buffer each."""
    lst = _side_streams.setdefault(dev, [])
    while len(lst) < n:
        lst.append(torch.cuda.Stream(device=dev))
    return lst[:n]


def candidate_multipliers(eq_alpha, eq_beta, eq_n, dev):
    """Reference quant_layers/linear.py:544: python-float grid rounded to fp32 (eq_n+1 entries)."""
    key = (float(eq_alpha), float(eq_beta), int(eq_n), str(dev))
    t = _mult_cache.get(key)
    if t is None:
        t = torch.tensor([eq_alpha + i * (eq_beta - eq_alpha) / eq_n for i in range(eq_n + 1)],
                         dtype=torch.float32).to(dev)
        _mult_cache[key] = t
    return t


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def stream_ptr(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def metric_id(name):
    if name not in _lib.METRICS:
        raise NotImplementedError(f"metric {name} not implemented!")
    return _lib.METRICS[name]


def linear_calibrate(*, weight, bias, x, out, grad, w_bit, a_bit, metric, eq_alpha, eq_beta, eq_n, search_round,
                     n_V, n_H, n_a, init_layerwise=False, postgelu=False, want_scores=False, force_f32=False, memoize=True):
    """Run calibration_step2 of a (post-GELU) Linear on the GPU.  Returns (w_interval[n_V*n_H], a_interval[n_a], scores, best)."""
    lib = _lib.load()
    dev = device_of(x, weight)
    weight, bias, x, out, grad = (to_dev(t, dev) for t in (weight, bias, x, out, grad))
    x = x.contiguous(); out = out.contiguous(); weight = weight.contiguous()
    grad = grad.contiguous() if grad is not None else None
    batch = x.shape[0]
    K = x.shape[-1]
    tokens = x.numel() // (batch * K)
    N = weight.shape[0]
    d = _lib.LinearDesc(batch, tokens, K, N, n_V, n_H, n_a, w_bit, a_bit, metric_id(metric), eq_n, search_round,
                        int(postgelu), int(init_layerwise), int(bias is not None), int(force_f32) | (0 if memoize else 2))
    need = lib.p4v_linear_workspace_bytes(C.byref(d))
    if need == 0:
        _lib.check(-1 if not lib.p4v_last_error() else -2, "p4v_linear_workspace_bytes")
    ws = workspace(dev, need)
    mult = candidate_multipliers(eq_alpha, eq_beta, eq_n, dev)
    w_iv = torch.empty(n_V * n_H, dtype=torch.float32, device=dev)
    a_iv = torch.empty(n_a, dtype=torch.float32, device=dev)
    scores = torch.zeros(search_round, 2, eq_n, n_V, dtype=torch.float32, device=dev) if want_scores else None
    best = torch.zeros(search_round, 2, n_V, dtype=torch.int32, device=dev) if want_scores else None
    with torch.cuda.device(dev):
        rc = lib.p4v_linear_calibrate(C.byref(d), ptr(weight), ptr(bias), ptr(x), ptr(out), ptr(grad), ptr(mult),
                                      ptr(w_iv), ptr(a_iv), ptr(scores), ptr(best), ptr(ws), ws.numel(), stream_ptr(dev))
    _lib.check(rc, "p4v_linear_calibrate")
    return w_iv, a_iv, scores, best


def linear_quant_forward(*, weight, bias, x, w_interval, a_interval, w_bit, a_bit, n_V, n_H, n_a, postgelu=False):
    """quant_forward of a calibrated (post-GELU) Linear as one int8 MFMA GEMM (p4v_linear_quant_forward)."""
    lib = _lib.load()
    dev = device_of(x, weight)
    weight, bias, x = (to_dev(t, dev) for t in (weight, bias, x))
    x = x.contiguous(); weight = weight.contiguous()
    batch, K = x.shape[0], x.shape[-1]
    tokens = x.numel() // (batch * K)
    N = weight.shape[0]
    d = _lib.LinearDesc(batch, tokens, K, N, n_V, n_H, n_a, w_bit, a_bit, metric_id("L2_norm"), 1, 1,
                        int(postgelu), 0, int(bias is not None), 4)
    need = lib.p4v_linear_workspace_bytes(C.byref(d))
    if need == 0:
        _lib.check(-2, "p4v_linear_workspace_bytes")
    ws = workspace(dev, need)
    w_iv = to_dev(w_interval, dev).reshape(-1).contiguous()
    a_iv = to_dev(a_interval, dev).reshape(-1).contiguous()
    out = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.p4v_linear_quant_forward(C.byref(d), ptr(weight), ptr(bias), ptr(x), ptr(w_iv), ptr(a_iv), ptr(out),
                                          ptr(ws), ws.numel(), stream_ptr(dev))
    _lib.check(rc, "p4v_linear_quant_forward")
    return out


def matmul_quant_forward(*, A, B, A_interval, B_interval, split, A_bit, B_bit, sos=False):
    """quant_forward of a calibrated MatMul (head-wise intervals; optional split-of-softmax twin on A)."""
    lib = _lib.load()
    dev = device_of(A, B)
    A, B = to_dev(A, dev), to_dev(B, dev)
    b, H, M, K = A.shape
    N = B.shape[3]
    d = _lib.MatMulDesc()
    d.batch, d.heads, d.M, d.K, d.N = b, H, M, K, N
    for i in range(4):
        d.a_stride[i] = A.stride(i)
        d.b_stride[i] = B.stride(i)
    d.A_bit, d.B_bit, d.metric, d.eq_n, d.search_round = A_bit, B_bit, metric_id("L2_norm"), 1, 1
    d.sos, d.init_layerwise, d.reserved = int(sos), 0, 4
    need = lib.p4v_matmul_workspace_bytes(C.byref(d))
    if need == 0:
        _lib.check(-2, "p4v_matmul_workspace_bytes")
    ws = workspace(dev, need)
    A_iv = to_dev(A_interval, dev).reshape(-1).contiguous()
    B_iv = to_dev(B_interval, dev).reshape(-1).contiguous()
    sp = to_dev(split, dev).reshape(-1).contiguous() if sos else None
    out = torch.empty(b, H, M, N, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.p4v_matmul_quant_forward(C.byref(d), ptr(A), ptr(B), ptr(A_iv), ptr(B_iv), ptr(sp), ptr(out),
                                          ptr(ws), ws.numel(), stream_ptr(dev))
    _lib.check(rc, "p4v_matmul_quant_forward")
    return out


def matmul_calibrate(*, A, B, out, grad, A_bit, B_bit, metric, eq_alpha, eq_beta, eq_n, search_round,
                     sos=False, init_layerwise=False, want_scores=False):
    """Run calibration_step2 of a MatMul (head-wise; optional split-of-softmax on A) on the GPU."""
    lib = _lib.load()
    dev = device_of(A, B)
    A, B, out, grad = (to_dev(t, dev) for t in (A, B, out, grad))
    out = out.contiguous()
    grad = grad.contiguous() if grad is not None else None
    b, H, M, K = A.shape
    N = B.shape[3]
    d = _lib.MatMulDesc()
    d.batch, d.heads, d.M, d.K, d.N = b, H, M, K, N
    for i in range(4):
        d.a_stride[i] = A.stride(i)
        d.b_stride[i] = B.stride(i)
    d.A_bit, d.B_bit, d.metric, d.eq_n, d.search_round = A_bit, B_bit, metric_id(metric), eq_n, search_round
    d.sos, d.init_layerwise, d.reserved = int(sos), int(init_layerwise), 0
    need = lib.p4v_matmul_workspace_bytes(C.byref(d))
    if need == 0:
        _lib.check(-2, "p4v_matmul_workspace_bytes")
    ws = workspace(dev, need)
    mult = candidate_multipliers(eq_alpha, eq_beta, eq_n, dev)
    A_iv = torch.empty(1 if sos else H, dtype=torch.float32, device=dev)
    B_iv = torch.empty(H, dtype=torch.float32, device=dev)
    split = torch.empty(1, dtype=torch.float32, device=dev) if sos else None
    scores = torch.zeros(search_round, 2, eq_n, H, dtype=torch.float32, device=dev) if want_scores else None
    best = torch.zeros(search_round, 2, H, dtype=torch.int32, device=dev) if want_scores else None
    with torch.cuda.device(dev):
        rc = lib.p4v_matmul_calibrate(C.byref(d), ptr(A), ptr(B), ptr(out), ptr(grad), ptr(mult), ptr(A_iv), ptr(B_iv),
                                      ptr(split), ptr(scores), ptr(best), ptr(ws), ws.numel(), stream_ptr(dev))
    _lib.check(rc, "p4v_matmul_calibrate")
    return A_iv, B_iv, split, scores, best


def conv_calibrate(*, weight, bias, x, out, grad, stride, padding, dilation, w_bit, a_bit, metric, eq_alpha, eq_beta,
                   eq_n, search_round, channelwise=True, init_layerwise=False, want_scores=False):
    """Run calibration_step2 of the patch-embedding Conv2d on the GPU."""
    lib = _lib.load()
    dev = device_of(x, weight)
    weight, bias, x, out, grad = (to_dev(t, dev) for t in (weight, bias, x, out, grad))
    weight = weight.contiguous(); x = x.contiguous(); out = out.contiguous()
    grad = grad.contiguous() if grad is not None else None
    b, ic, H, W = x.shape
    oc, _, kh, kw = weight.shape
    d = _lib.ConvDesc(b, ic, H, W, oc, kh, kw, stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1],
                      w_bit, a_bit, metric_id(metric), eq_n, search_round, int(channelwise), int(init_layerwise),
                      int(bias is not None), 0)
    need = lib.p4v_conv_workspace_bytes(C.byref(d))
    if need == 0:
        _lib.check(-2, "p4v_conv_workspace_bytes")
    ws = workspace(dev, need)
    mult = candidate_multipliers(eq_alpha, eq_beta, eq_n, dev)
    nw = oc if channelwise else 1
    w_iv = torch.empty(nw, dtype=torch.float32, device=dev)
    a_iv = torch.empty(1, dtype=torch.float32, device=dev)
    scores = torch.zeros(search_round, 2, eq_n, nw, dtype=torch.float32, device=dev) if want_scores else None
    best = torch.zeros(search_round, 2, nw, dtype=torch.int32, device=dev) if want_scores else None
    with torch.cuda.device(dev):
        rc = lib.p4v_conv_calibrate(C.byref(d), ptr(weight), ptr(bias), ptr(x), ptr(out), ptr(grad), ptr(mult),
                                    ptr(w_iv), ptr(a_iv), ptr(scores), ptr(best), ptr(ws), ws.numel(), stream_ptr(dev))
    _lib.check(rc, "p4v_conv_calibrate")
    return w_iv, a_iv, scores, best


def quantize_i8(x2d, scales, rows_per_scale, lo, hi):
    """int8 grid indices clamp(rint(x/s), lo, hi) of a 2-D fp32 tensor (K padded to 64 with zeros)."""
    lib = _lib.load()
    _require_cuda(x2d, "x")
    x2d = x2d.contiguous().float()
    rows, cols = x2d.shape
    colsp = (cols + 63) // 64 * 64
    q = torch.empty(rows, colsp, dtype=torch.int8, device=x2d.device)
    scales = scales.to(x2d.device, torch.float32).contiguous()
    rc = lib.p4v_quantize_i8(ptr(x2d), rows, cols, colsp, ptr(scales), rows_per_scale, lo, hi, ptr(q), stream_ptr(x2d.device))
    _lib.check(rc, "p4v_quantize_i8")
    return q[:, :cols]


def stats_enable(flag):
    _lib.load().p4v_stats_enable(int(flag))


def stats_reset():
    _lib.load().p4v_stats_reset()


def stats_get():
    s = _lib.KernelStats()
    _lib.check(_lib.load().p4v_stats_get(C.byref(s)), "p4v_stats_get")
    return {k: getattr(s, k) for k, _ in _lib.KernelStats._fields_}
