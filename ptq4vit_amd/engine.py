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


def workspace(dev, nbytes, slot=0):
    """Scratch for one calibration call; one buffer per (device, stream, member of a group) so that searches running on
    different streams, and the members of a p4v_calibrate_group call, never share scratch."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream, slot)
    ws = _workspace.get(key)
    if ws is None or ws.numel() < nbytes:
        _workspace.pop(key, None)
        ws = None           # (the old buffer goes back to torch's pool BEFORE the new one is taken from it)
        pad = (1 << 20) if slot == "arena" else int(nbytes * 0.05) + (1 << 20)    # an arena is sized by its caller's budget
        ws = torch.empty(int(nbytes) + pad, dtype=torch.uint8, device=dev)
        _workspace[key] = ws
    return ws


def release_workspace():
    _workspace.clear()


def workspace_bytes(dev=None):
    """Bytes of scratch buffers the engine keeps alive between calls (per device, stream and group member)."""
    return sum(ws.numel() for (d, _s, _slot), ws in _workspace.items() if dev is None or d == dev)


def workspace_held(dev, slot=0):
    """Size of the scratch buffer already kept for (current stream of `dev`, `slot`); 0 if none."""
    ws = _workspace.get((dev, torch.cuda.current_stream(dev).cuda_stream, slot))
    return 0 if ws is None else ws.numel()


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


class Job:
    """One prepared p4v_*_calibrate call: descriptor, device tensors (kept alive until the results are taken), outputs,
    workspace size.  `run_job` makes the call; `calibrate_group` makes the calls of many jobs as ONE p4v_calibrate_group."""
    __slots__ = ("kind", "name", "desc", "inputs", "mult", "outputs", "need", "dev", "scores", "best")

    def __init__(self, kind, name, desc, inputs, mult, outputs, need, dev, scores=None, best=None):
        self.kind, self.name, self.desc, self.inputs, self.mult = kind, name, desc, inputs, mult
        self.outputs, self.need, self.dev, self.scores, self.best = outputs, need, dev, scores, best


def _need(fn, desc, what):
    need = fn(C.byref(desc))
    if need == 0:
        _lib.check(-1 if not _lib.load().p4v_last_error() else -2, what)
    return need


def run_job(job):
    """The single-module call (p4v_linear_calibrate / p4v_matmul_calibrate / p4v_conv_calibrate) on the current stream."""
    lib = _lib.load()
    ws = workspace(job.dev, job.need)
    with torch.cuda.device(job.dev):
        rc = getattr(lib, job.name)(C.byref(job.desc), *[ptr(t) for t in job.inputs], ptr(job.mult), *[ptr(t) for t in job.outputs],
                                    ptr(job.scores), ptr(job.best), ptr(ws), ws.numel(), stream_ptr(job.dev))
    _lib.check(rc, job.name)
    return job


def calibrate_group(jobs, inputs_ready=None):
    """calibration_step2 of all `jobs` in ONE p4v_calibrate_group call on the current stream: the members search in lock
    step, every kernel launch of the same kind is issued once for all of them.  Results bit-identical to run_job on each.
    `inputs_ready`: a torch.cuda.Event recorded behind the capture passes that fill the jobs' captured tensors -- the call
    starts at once with the work that needs none of them (weight abs-max, candidate tables, the candidate planes of the
    weights) and makes its stream wait for the event where the first captured tensor is read.  None: the tensors are ready."""
    jobs = list(jobs)
    if not jobs:
        return jobs
    lib = _lib.load()
    dev = jobs[0].dev
    arr = (_lib.GroupJob * len(jobs))()
    # ONE arena per (device, stream) for the scratch of all members, carved by offset: the call's scratch is exactly the sum of
    # the members' needs whatever mixture the previous calls on this stream held (a buffer per member slot, each grown to the
    # largest member it ever saw, added up to more than the budget of a 128-image configuration and fragmented torch's pool).
    offs, total = [], 0
    for job in jobs:
        offs.append(total)
        total += (int(job.need) + 4095) & ~4095
    arena = workspace(dev, total, slot="arena")
    base = arena.data_ptr()
    keep = [arena]
    for i, job in enumerate(jobs):
        if job.dev != dev:
            raise ValueError("calibrate_group: every member must live on the same device")
        if job.scores is not None:
            raise ValueError("calibrate_group: score tables are only returned by the single-module calls")
        g = arr[i]
        g.kind, g.status, g.desc = job.kind, 0, C.cast(C.pointer(job.desc), C.c_void_p)
        ins = list(job.inputs) + [None] * (5 - len(job.inputs))
        for k in range(5):
            g.inp[k] = ins[k].data_ptr() if ins[k] is not None else None
        g.mult = job.mult.data_ptr()
        outs = list(job.outputs) + [None] * (3 - len(job.outputs))
        for k in range(3):
            g.out[k] = outs[k].data_ptr() if outs[k] is not None else None
        g.workspace, g.workspace_bytes = base + offs[i], int(job.need)
    with torch.cuda.device(dev):
        ev = C.c_void_p(inputs_ready.cuda_event) if inputs_ready is not None else C.c_void_p(0)
        rc = lib.p4v_calibrate_group(arr, len(jobs), stream_ptr(dev), ev)
    _lib.check(rc, "p4v_calibrate_group")
    del keep
    return jobs


def launch_counters(reset=False):
    """Process-wide launch counters (p4v_launch_counters): launches the calibration path asked for, launches issued to the
    GPU (a grouped launch counts once), issue rounds of the groups, group calls."""
    out = (C.c_int64 * 4)()
    _lib.check(_lib.load().p4v_launch_counters(out, int(bool(reset))), "p4v_launch_counters")
    return dict(zip(("asked", "issued", "rounds", "groups"), (int(v) for v in out)))


def linear_job(*, weight, bias, x, out, grad, w_bit, a_bit, metric, eq_alpha, eq_beta, eq_n, search_round,
               n_V, n_H, n_a, init_layerwise=False, postgelu=False, want_scores=False, force_f32=False, memoize=True,
               prune=True):
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
                        int(postgelu), int(init_layerwise), int(bias is not None),
                        int(force_f32) | (0 if memoize else 2) | (0 if prune else 8))
    need = _need(lib.p4v_linear_workspace_bytes, d, "p4v_linear_workspace_bytes")
    mult = candidate_multipliers(eq_alpha, eq_beta, eq_n, dev)
    w_iv = torch.empty(n_V * n_H, dtype=torch.float32, device=dev)
    a_iv = torch.empty(n_a, dtype=torch.float32, device=dev)
    scores = torch.zeros(search_round, 2, eq_n, n_V, dtype=torch.float32, device=dev) if want_scores else None
    best = torch.zeros(search_round, 2, n_V, dtype=torch.int32, device=dev) if want_scores else None
    return Job(_lib.JOB_LINEAR, "p4v_linear_calibrate", d, (weight, bias, x, out, grad), mult, (w_iv, a_iv), need, dev, scores, best)


def linear_calibrate(**kw):
    """Run calibration_step2 of a (post-GELU) Linear on the GPU.  Returns (w_interval[n_V*n_H], a_interval[n_a], scores, best)."""
    job = run_job(linear_job(**kw))
    return job.outputs[0], job.outputs[1], job.scores, job.best


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


def matmul_job(*, A, B, out, grad, A_bit, B_bit, metric, eq_alpha, eq_beta, eq_n, search_round,
               sos=False, init_layerwise=False, want_scores=False, prune=True):
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
    d.sos, d.init_layerwise, d.reserved = int(sos), int(init_layerwise), (0 if prune else 8)
    need = _need(lib.p4v_matmul_workspace_bytes, d, "p4v_matmul_workspace_bytes")
    mult = candidate_multipliers(eq_alpha, eq_beta, eq_n, dev)
    A_iv = torch.empty(1 if sos else H, dtype=torch.float32, device=dev)
    B_iv = torch.empty(H, dtype=torch.float32, device=dev)
    split = torch.empty(1, dtype=torch.float32, device=dev) if sos else None
    scores = torch.zeros(search_round, 2, eq_n, H, dtype=torch.float32, device=dev) if want_scores else None
    best = torch.zeros(search_round, 2, H, dtype=torch.int32, device=dev) if want_scores else None
    return Job(_lib.JOB_MATMUL, "p4v_matmul_calibrate", d, (A, B, out, grad), mult, (A_iv, B_iv, split), need, dev, scores, best)


def matmul_calibrate(**kw):
    """Run calibration_step2 of a MatMul (head-wise; optional split-of-softmax on A) on the GPU."""
    job = run_job(matmul_job(**kw))
    return job.outputs[0], job.outputs[1], job.outputs[2], job.scores, job.best


def conv_job(*, weight, bias, x, out, grad, stride, padding, dilation, w_bit, a_bit, metric, eq_alpha, eq_beta,
             eq_n, search_round, channelwise=True, init_layerwise=False, want_scores=False, prune=True):
    lib = _lib.load()
    dev = device_of(x, weight)
    weight, bias, x, out, grad = (to_dev(t, dev) for t in (weight, bias, x, out, grad))
    weight = weight.contiguous(); x = x.contiguous(); out = out.contiguous()
    grad = grad.contiguous() if grad is not None else None
    b, ic, H, W = x.shape
    oc, _, kh, kw = weight.shape
    d = _lib.ConvDesc(b, ic, H, W, oc, kh, kw, stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1],
                      w_bit, a_bit, metric_id(metric), eq_n, search_round, int(channelwise), int(init_layerwise),
                      int(bias is not None), 0 if prune else 8)
    need = _need(lib.p4v_conv_workspace_bytes, d, "p4v_conv_workspace_bytes")
    mult = candidate_multipliers(eq_alpha, eq_beta, eq_n, dev)
    nw = oc if channelwise else 1
    w_iv = torch.empty(nw, dtype=torch.float32, device=dev)
    a_iv = torch.empty(1, dtype=torch.float32, device=dev)
    scores = torch.zeros(search_round, 2, eq_n, nw, dtype=torch.float32, device=dev) if want_scores else None
    best = torch.zeros(search_round, 2, nw, dtype=torch.int32, device=dev) if want_scores else None
    return Job(_lib.JOB_CONV, "p4v_conv_calibrate", d, (weight, bias, x, out, grad), mult, (w_iv, a_iv), need, dev, scores, best)


def conv_calibrate(**kw):
    """Run calibration_step2 of the patch-embedding Conv2d on the GPU (prune=False: no exact candidate pruning)."""
    job = run_job(conv_job(**kw))
    return job.outputs[0], job.outputs[1], job.scores, job.best


# ----------------------------------------------------------------------------------------------------------------
# Granular passes: one part of calibration_step2 per call (C ABI p4v_amax_init_* / p4v_*_search_*), for callers that
# drive the alternation themselves like the reference's _initialize_intervals / _search_best_*_interval methods.
# ----------------------------------------------------------------------------------------------------------------
def _flat(t, dev, rows=None):
    t = to_dev(t, dev).contiguous()
    return t.reshape(rows, -1) if rows is not None else t.reshape(-1)


class _Stepper:
    """Tensors + descriptor + scratch of one module, shared by the granular calls on it."""

    def _call(self, name, *args):
        with torch.cuda.device(self.dev):
            rc = getattr(self.lib, name)(C.byref(self.d), *[ptr(a) for a in args], ptr(self.ws), self.ws.numel(),
                                         stream_ptr(self.dev))
        _lib.check(rc, name)

    def _tables(self, want, blocks, eq_n=None):
        eq_n = self.d.eq_n if eq_n is None else eq_n
        if not want:
            return None, None
        return (torch.zeros(eq_n, blocks, dtype=torch.float32, device=self.dev),
                torch.zeros(blocks, dtype=torch.int32, device=self.dev))


class LinearStepper(_Stepper):
    def __init__(self, *, weight, bias, x, out, grad, w_bit, a_bit, metric, eq_n, n_V, n_H, n_a, init_layerwise=False,
                 postgelu=False, force_f32=False):
        self.lib = _lib.load()
        self.dev = dev = device_of(x, weight)
        self.weight, self.bias, self.x, self.out, self.grad = (
            (to_dev(t, dev).contiguous() if t is not None else None) for t in (weight, bias, x, out, grad))
        batch, K = self.x.shape[0], self.x.shape[-1]
        self.blocks_w, self.blocks_a, self.n_V = n_V * n_H, n_a, n_V
        self.d = _lib.LinearDesc(batch, self.x.numel() // (batch * K), K, self.weight.shape[0], n_V, n_H, n_a, w_bit, a_bit,
                                 metric_id(metric), eq_n, 1, int(postgelu), int(init_layerwise), int(bias is not None),
                                 int(force_f32))
        need = self.lib.p4v_linear_workspace_bytes(C.byref(self.d))
        if need == 0:
            _lib.check(-2, "p4v_linear_workspace_bytes")
        self.ws = workspace(dev, need)

    def init_intervals(self):
        w_iv = torch.empty(self.blocks_w, dtype=torch.float32, device=self.dev)
        a_iv = torch.empty(self.blocks_a, dtype=torch.float32, device=self.dev)
        self._call("p4v_amax_init_linear", self.weight, self.x, w_iv, a_iv)
        return w_iv, a_iv

    def search_w(self, w_cands, w_interval, a_interval, want_scores=False):
        """Returns (new w_interval [n_V*n_H], scores [eq_n][n_V] | None, best [n_V] | None)."""
        w_iv = _flat(w_interval, self.dev).clone()
        scores, best = self._tables(want_scores, self.n_V)
        self._call("p4v_linear_search_w", self.weight, self.bias, self.x, self.out, self.grad,
                   _flat(w_cands, self.dev, self.d.eq_n + 1), w_iv, _flat(a_interval, self.dev), scores, best)
        return w_iv, scores, best

    def search_a(self, a_cands, w_interval, a_interval, want_scores=False):
        a_iv = _flat(a_interval, self.dev).clone()
        scores, best = self._tables(want_scores, self.n_V)
        self._call("p4v_linear_search_a", self.weight, self.bias, self.x, self.out, self.grad,
                   _flat(a_cands, self.dev, self.d.eq_n + 1), _flat(w_interval, self.dev), a_iv, scores, best)
        return a_iv, scores, best


class MatMulStepper(_Stepper):
    def __init__(self, *, A, B, out, grad, A_bit, B_bit, metric, eq_n, sos=False, init_layerwise=False):
        self.lib = _lib.load()
        self.dev = dev = device_of(A, B)
        self.A, self.B = to_dev(A, dev), to_dev(B, dev)
        self.out = to_dev(out, dev).contiguous() if out is not None else None
        self.grad = to_dev(grad, dev).contiguous() if grad is not None else None
        b, H, M, K = self.A.shape
        self.H, self.sos = H, bool(sos)
        d = self.d = _lib.MatMulDesc()
        d.batch, d.heads, d.M, d.K, d.N = b, H, M, K, self.B.shape[3]
        for i in range(4):
            d.a_stride[i] = self.A.stride(i)
            d.b_stride[i] = self.B.stride(i)
        d.A_bit, d.B_bit, d.metric, d.eq_n, d.search_round = A_bit, B_bit, metric_id(metric), eq_n, 1
        d.sos, d.init_layerwise, d.reserved = int(sos), int(init_layerwise), 0
        need = self.lib.p4v_matmul_workspace_bytes(C.byref(d))
        if need == 0:
            _lib.check(-2, "p4v_matmul_workspace_bytes")
        self.ws = workspace(dev, need)

    def init_intervals(self):
        """(A_interval [heads] -- None for the split-of-softmax class, whose split search sets it -- , B_interval [heads])"""
        A_iv = torch.empty(self.H, dtype=torch.float32, device=self.dev)
        B_iv = torch.empty(self.H, dtype=torch.float32, device=self.dev)
        self._call("p4v_amax_init_matmul", self.A, self.B, A_iv, B_iv)
        return (None if self.sos else A_iv), B_iv

    def search_A(self, A_cands, A_interval, B_interval, want_scores=False):
        A_iv = _flat(A_interval, self.dev).clone()
        scores, best = self._tables(want_scores, self.H)
        self._call("p4v_matmul_search_A", self.A, self.B, self.out, self.grad, _flat(A_cands, self.dev, self.d.eq_n + 1),
                   A_iv, _flat(B_interval, self.dev), scores, best)
        return A_iv, scores, best

    def search_split(self, want_scores=False):
        """SoS: (split [1], A_interval [1] = split/(qmax-1), scores [20][1] | None, best [1] | None)."""
        split = torch.empty(1, dtype=torch.float32, device=self.dev)
        A_iv = torch.empty(1, dtype=torch.float32, device=self.dev)
        scores = best = None
        if want_scores:
            if self.d.eq_n < 20:
                raise ValueError("the split score table has 20 rows: eq_n >= 20 needed to return it")
            scores, best = self._tables(True, self.H)
        self._call("p4v_sos_search_split", self.A, self.B, self.out, self.grad, split, A_iv, scores, best)
        return split, A_iv, (scores[:20, :1] if scores is not None else None), (best[:1] if best is not None else None)

    def search_B(self, B_cands, A_interval, B_interval, split=None, want_scores=False):
        B_iv = _flat(B_interval, self.dev).clone()
        scores, best = self._tables(want_scores, self.H)
        self._call("p4v_matmul_search_B", self.A, self.B, self.out, self.grad, _flat(B_cands, self.dev, self.d.eq_n + 1),
                   _flat(A_interval, self.dev), (_flat(split, self.dev) if self.sos else None), B_iv, scores, best)
        return B_iv, scores, best


class ConvStepper(_Stepper):
    def __init__(self, *, weight, bias, x, out, grad, stride, padding, dilation, w_bit, a_bit, metric, eq_n,
                 channelwise=True, init_layerwise=False):
        self.lib = _lib.load()
        self.dev = dev = device_of(x, weight)
        self.weight, self.bias, self.x, self.out, self.grad = (
            (to_dev(t, dev).contiguous() if t is not None else None) for t in (weight, bias, x, out, grad))
        b, ic, H, W = self.x.shape
        oc, _, kh, kw = self.weight.shape
        self.nw, self.channelwise = (oc if channelwise else 1), bool(channelwise)
        self.d = _lib.ConvDesc(b, ic, H, W, oc, kh, kw, stride[0], stride[1], padding[0], padding[1], dilation[0],
                               dilation[1], w_bit, a_bit, metric_id(metric), eq_n, 1, int(channelwise),
                               int(init_layerwise), int(bias is not None), 0)
        need = self.lib.p4v_conv_workspace_bytes(C.byref(self.d))
        if need == 0:
            _lib.check(-2, "p4v_conv_workspace_bytes")
        self.ws = workspace(dev, need)

    def init_intervals(self):
        w_iv = torch.empty(self.nw, dtype=torch.float32, device=self.dev)
        a_iv = torch.empty(1, dtype=torch.float32, device=self.dev)
        self._call("p4v_amax_init_conv", self.weight, self.x, w_iv, a_iv)
        return w_iv, a_iv

    def search_w(self, w_cands, w_interval, a_interval, want_scores=False):
        w_iv = _flat(w_interval, self.dev).clone()
        scores, best = self._tables(want_scores, self.nw)
        name = "p4v_conv_search_w_channelwise" if self.channelwise else "p4v_conv_search_w_layerwise"
        self._call(name, self.weight, self.bias, self.x, self.out, self.grad, _flat(w_cands, self.dev, self.d.eq_n + 1),
                   w_iv, _flat(a_interval, self.dev), scores, best)
        return w_iv, scores, best

    def search_a(self, a_cands, w_interval, a_interval, want_scores=False):
        a_iv = _flat(a_interval, self.dev).clone()
        scores, best = self._tables(want_scores, self.nw)
        self._call("p4v_conv_search_a", self.weight, self.bias, self.x, self.out, self.grad,
                   _flat(a_cands, self.dev, self.d.eq_n + 1), _flat(w_interval, self.dev), a_iv, scores, best)
        return a_iv, scores, best


def score_argmax_gather(scores, cands):
    """(interval [blocks], best [blocks]): first-maximum argmax over the candidate axis, NaN counts as the maximum."""
    lib = _lib.load()
    _require_cuda(scores, "scores")
    dev = scores.device
    scores = scores.float().contiguous()
    eq_n, blocks = scores.shape
    cands = to_dev(cands, dev).contiguous().reshape(-1, blocks)
    if cands.shape[0] < eq_n:
        raise ValueError("candidate table shorter than the score table")
    iv = torch.empty(blocks, dtype=torch.float32, device=dev)
    best = torch.empty(blocks, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.p4v_score_argmax_gather(ptr(scores), eq_n, blocks, ptr(cands), ptr(iv), ptr(best), stream_ptr(dev))
    _lib.check(rc, "p4v_score_argmax_gather")
    return iv, best


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


def pack_plane_i8(x2d, *, mode="sym", scales=None, rows_per_scale=1, lo=-128, hi=127, qmax=128, const_scale=0.0):
    """ONE int8 operand plane exactly as the candidate sweeps consume it (p4v_pack_plane_i8): mode "sym"
    (clamp(rint(x/s), lo, hi); `scales=None` uses `const_scale`, the post-GELU negative range), "sos_hi" / "sos_lo"
    (split-of-softmax ranges, `scales` = the split), "twin" (both post-GELU ranges in one plane: `scales` = the positive
    interval, `const_scale` the negative one, lo < 0 < hi).  Returns the [rows][cols] int8 plane (padding stripped)."""
    lib = _lib.load()
    _require_cuda(x2d, "x")
    x2d = x2d.contiguous().float()
    rows, cols = x2d.shape
    colsp = (cols + 63) // 64 * 64
    q = torch.empty(rows, colsp, dtype=torch.int8, device=x2d.device)
    d = _lib.PlaneDesc(rows, cols, colsp, int(rows_per_scale),
                       {"sym": _lib.PLANE_SYM, "sos_hi": _lib.PLANE_SOS_HI, "sos_lo": _lib.PLANE_SOS_LO, "twin": _lib.PLANE_TWIN}[mode],
                       int(lo), int(hi), int(qmax), float(const_scale), 0)
    sc = scales.to(x2d.device, torch.float32).reshape(-1).contiguous() if scales is not None else None
    with torch.cuda.device(x2d.device):
        rc = lib.p4v_pack_plane_i8(C.byref(d), ptr(x2d), ptr(sc), ptr(q), stream_ptr(x2d.device))
    _lib.check(rc, "p4v_pack_plane_i8")
    return q[:, :cols], q


def fake_quant(x2d, scales, rows_per_scale, lo, hi):
    """clamp(rint(x / s), lo, hi) * s of a 2-D fp32 tensor, s = scales[row // rows_per_scale] (p4v_fake_quant)."""
    lib = _lib.load()
    _require_cuda(x2d, "x")
    x2d = x2d.contiguous().float()
    rows, cols = x2d.shape
    y = torch.empty_like(x2d)
    scales = scales.to(x2d.device, torch.float32).reshape(-1).contiguous()
    with torch.cuda.device(x2d.device):
        rc = lib.p4v_fake_quant(ptr(x2d), rows, cols, ptr(scales), int(rows_per_scale), int(lo), int(hi), ptr(y),
                                stream_ptr(x2d.device))
    _lib.check(rc, "p4v_fake_quant")
    return y


def _pad4(seq, fill):
    seq = list(seq)
    return [fill] * (4 - len(seq)) + seq


def export_quantize(src, *, mode, scale1, scale1_stride, scale1_div, lo1=0, hi1=0, scale2=None, scale2_stride=(0, 0, 0, 0),
                    scale2_div=(1, 1, 1, 1), scale2_const=0.0, lo2=0, hi2=0, qmax=128, dims=None, src_stride=None):
    """Integer image of `src` on the GPU (p4v_export_quantize).  `src` is viewed as the 4-D tensor `dims` through the
    element strides `src_stride` (default: src's own shape / strides, left-padded to 4-D).  Returns a contiguous device
    tensor of `dims`: int8 (sym_i8), float32 (sym_f32) or uint8 (gelu_u8 / sos_u8)."""
    lib = _lib.load()
    _require_cuda(src, "export source")
    dev = src.device
    src = src.detach()
    if src.dtype != torch.float32:
        if src_stride is not None and not src.is_contiguous():
            # the caller's strides describe `src` as it is; `.float()` of a non-dense view has other strides
            raise TypeError("export_quantize: explicit src_stride needs a float32 source (convert before deriving the strides)")
        src = src.float()
    if dims is None:
        dims, src_stride = _pad4(src.shape, 1), _pad4(src.stride(), 0)
    code = {"sym_i8": _lib.EXPORT_SYM_I8, "sym_f32": _lib.EXPORT_SYM_F32, "gelu_u8": _lib.EXPORT_GELU_U8,
            "sos_u8": _lib.EXPORT_SOS_U8}[mode]
    out = torch.empty(tuple(dims), dtype={"sym_i8": torch.int8, "sym_f32": torch.float32}.get(mode, torch.uint8), device=dev)
    d = _lib.ExportDesc()
    for i in range(4):
        d.dims[i], d.src_stride[i] = int(dims[i]), int(src_stride[i])
        d.scale1_stride[i], d.scale1_div[i] = int(scale1_stride[i]), int(scale1_div[i])
        d.scale2_stride[i], d.scale2_div[i] = int(scale2_stride[i]), int(scale2_div[i])
    d.scale2_const = float(scale2_const)
    d.mode, d.lo1, d.hi1, d.lo2, d.hi2, d.qmax, d.reserved = code, int(lo1), int(hi1), int(lo2), int(hi2), int(qmax), 0
    s1 = to_dev(torch.as_tensor(scale1), dev).reshape(-1).contiguous()
    s2 = to_dev(torch.as_tensor(scale2), dev).reshape(-1).contiguous() if scale2 is not None else None
    with torch.cuda.device(dev):
        rc = lib.p4v_export_quantize(C.byref(d), ptr(src), ptr(s1), ptr(s2), ptr(out), stream_ptr(dev))
    _lib.check(rc, "p4v_export_quantize")
    return out


def multi_copy(table, n, index, max_bytes, dev):
    """One launch: for each of the `n` {src, dst base, bytes} triples of the device int64 `table`, copy `bytes` from src
    to dst base + index * bytes (p4v_multi_copy: the capture pass appends a sub-batch to every cache at once)."""
    lib = _lib.load()
    with torch.cuda.device(dev):
        rc = lib.p4v_multi_copy(ptr(table), int(n), int(index), int(max_bytes), stream_ptr(dev))
    _lib.check(rc, "p4v_multi_copy")


def stats_enable(flag):
    """Launch timing of the sweep kernels on the CALLING thread (bench.py roofline)."""
    _lib.load().p4v_stats_enable(int(bool(flag)))


def debug_variant(variant=0, force_generic=False):
    """Kernel A/B switches (tests, tools/bench_layer.py); 0 in production."""
    _lib.check(_lib.load().p4v_debug_set_variant(int(variant), int(bool(force_generic))), "p4v_debug_set_variant")


def debug_tuning(key, value):
    _lib.check(_lib.load().p4v_debug_set_tuning(int(key), int(value)), "p4v_debug_set_tuning")


def debug_topk_rows(mass, k):
    """Row selection of the exact pruning (k_topk_rows) on `mass` [segs, n] fp32: int32 [segs, k], the segment-local indices
    of the k heaviest entries in ascending order (ties: lowest indices).  For the tests."""
    assert mass.is_cuda and mass.dtype == torch.float32 and mass.dim() == 2 and mass.is_contiguous()
    segs, n = mass.shape
    out = torch.empty(segs, int(k), dtype=torch.int32, device=mass.device)
    with torch.cuda.device(mass.device):
        rc = _lib.load().p4v_debug_topk_rows(ptr(mass), segs, n, int(k), ptr(out), stream_ptr(mass.device))
    _lib.check(rc, "p4v_debug_topk_rows")
    return out


def stats_reset():
    _lib.load().p4v_stats_reset()


def stats_get():
    s = _lib.KernelStats()
    _lib.check(_lib.load().p4v_stats_get(C.byref(s)), "p4v_stats_get")
    return {k: getattr(s, k) for k, _ in _lib.KernelStats._fields_}


def prune_counters(reset=False):
    """Process-wide counters of the exact candidate pruning since the last reset (p4v_prune_counters): how many search
    passes ran in three stages, how many of those had no survivors besides the bound's candidates, how many eligible passes
    kept the full sweep, how many were never eligible (score tables, cosine, fp32 planes, pruning switched off)."""
    out = (C.c_int64 * 4)()
    _lib.check(_lib.load().p4v_prune_counters(out, int(bool(reset))), "p4v_prune_counters")
    return dict(zip(("staged", "staged_no_survivors", "kept_full_sweep", "not_eligible"), (int(v) for v in out)))


def stats_launches():
    """Every sweep launch the calling thread enqueued with timing enabled since the last stats_reset(), in launch order:
    [{"kernel", "stage", "grid_x", "grid_z", "ms", "ops", "alg_ops", "alg_bytes"}] (p4v_stats_launches)."""
    lib = _lib.load()
    n = C.c_int64(0)
    _lib.check(lib.p4v_stats_launches(None, 0, C.byref(n)), "p4v_stats_launches")
    buf = (_lib.LaunchRecord * max(1, n.value))()
    _lib.check(lib.p4v_stats_launches(buf, n.value, C.byref(n)), "p4v_stats_launches")
    return [{"kernel": _lib.LAUNCH_KINDS.get(r.kind, str(r.kind)), "stage": _lib.LAUNCH_STAGES.get(r.stage, str(r.stage)),
             "grid_x": r.grid_x, "grid_z": r.grid_z, "ms": r.ms, "ops": r.ops, "alg_ops": r.alg_ops, "alg_bytes": r.alg_bytes}
            for r in buf[:n.value]]
