"""Multi-threaded torch-CPU restatement of step 2 for the five hot classes.

TEST INFRASTRUCTURE ONLY, like ``oracle/ptq4vit_oracle.py``: imported by ``tests/`` and by ``bench.py``'s
``cpu_baseline`` leg (``kind: "port"``, ``port_backend: "torch"``), never by ``ptq4vit_amd``.

Why a second restatement: the numpy oracle is the parity checker -- written for clarity, its elementwise passes run on one
thread.  The reference's CPU path is torch: ``F.linear`` / ``@`` / ``F.conv2d`` on all host cores and multi-threaded
elementwise kernels.  This module restates the same algorithm with torch operators (same chunking idea: a few candidates
per GEMM, float32 throughout, IEEE division, round-half-even) so that the timed CPU baseline uses the host the way the
reference would.  It covers what the shipped configurations instantiate (n_H = n_a = 1, head-wise matmul intervals,
channel-wise / layer-wise patch embedding with a_bit = 32) and is validated against the golden files made by the reference
itself (tests/test_oracle_golden.py::test_torch_port_*): same selections (or near-ties by the reference's scores), intervals
bit-identical, score tables within SCORE_RTOL.  Reference lines are cited per function.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

POSTGELU_NEG_RANGE = 0.16997124254703522  # quant_layers/linear.py:574


def _mult(eq_alpha, eq_beta, eq_n):
    """quant_layers/linear.py:544: python-float grid -> float32."""
    return torch.tensor([eq_alpha + i * (eq_beta - eq_alpha) / eq_n for i in range(eq_n + 1)], dtype=torch.float32)


def _fq(x, s, lo, hi):
    """quant_layers/linear.py:167-168."""
    return torch.clamp(torch.round(x / s), lo, hi) * s


def _elementwise(raw, sim, metric, grad):
    """quant_layers/linear.py:409-421 (no reduction)."""
    d = raw - sim
    if metric == "L1_norm":
        return -d.abs()
    if metric == "L2_norm":
        return -(d * d)
    if metric == "linear_weighted_L2_norm":
        return -raw.abs() * d * d
    if metric == "square_weighted_L2_norm":
        return -((raw * d) ** 2)
    if metric == "hessian":
        return -((grad * d) ** 2)
    raise NotImplementedError(metric)


def _sim_last(raw, sim, metric, grad):
    """quant_layers/linear.py:399-424: similarity reduced over the last dim."""
    if metric == "cosine":
        return F.cosine_similarity(raw, sim, dim=-1)
    return _elementwise(raw, sim, metric, grad).mean(-1)


class TorchLinear:
    """PTQSLBatchingQuantLinear / PostGeluPTQSLBatchingQuantLinear step 2 (linear.py:349-642), n_H = n_a = 1.

    ``calibration_step2`` is the whole method; ``initial`` / ``score_w`` / ``score_a`` are its three pieces for callers
    that drive the alternation themselves (tests that follow another implementation's trajectory pass by pass)."""

    def __init__(self, weight, bias, *, w_bit=8, a_bit=8, metric="hessian", search_round=1, eq_alpha=0.0, eq_beta=1.0,
                 eq_n=100, n_V=1, postgelu=False, init_layerwise=False, chunk=10, **unused):
        self.w = torch.as_tensor(weight, dtype=torch.float32)
        self.b = None if bias is None else torch.as_tensor(bias, dtype=torch.float32)
        self.oc, self.ic = self.w.shape
        self.wq, self.aq = 2 ** (w_bit - 1), 2 ** (a_bit - 1)
        self.metric, self.R = metric, search_round
        self.mult, self.eq_n, self.n_V = _mult(eq_alpha, eq_beta, eq_n), eq_n, n_V
        self.postgelu, self.init_layerwise, self.chunk = postgelu, init_layerwise, chunk
        self.a_neg = POSTGELU_NEG_RANGE / self.aq
        self.trace = []

    def _qx(self, x, s):
        if not self.postgelu:
            return _fq(x, s, -self.aq, self.aq - 1)
        return _fq(x, s, 0, self.aq - 1) + _fq(x, self.a_neg, -self.aq, 0)          # linear.py:601-607

    def _reduce(self, sim, keep):
        """linear.py:483-487: mean over the token dims, sum over the batch."""
        mid = tuple(range(1, sim.dim() - keep))
        if mid:
            sim = sim.mean(dim=mid)
        return sim.sum(0)

    def initial(self, x):
        """linear.py:380-397 / 576-599 + :544-545: (w_interval [n_V], a_interval [], w candidates [eq_n+1, n_V], a candidates [eq_n+1])."""
        x = torch.as_tensor(x, dtype=torch.float32)
        nV, crb = self.n_V, self.oc // self.n_V
        wv = self.w.view(nV, crb, self.ic)
        if self.init_layerwise:
            w_iv = (self.w.abs().max() / (self.wq - 0.5)).repeat(nV)
        else:
            w_iv = wv.abs().amax(dim=(1, 2)) / (self.wq - 0.5)
        a_iv = (x.max() if self.postgelu else x.abs().max()) / (self.aq - 0.5)
        return w_iv, a_iv, self.mult[:, None] * w_iv[None, :], self.mult * a_iv

    def score_w(self, x, out, grad, w_c, a_iv):
        """linear.py:455-492: the (eq_n, n_V) table the reference feeds to argmax, activations quantised with `a_iv`."""
        nV, crb = self.n_V, self.oc // self.n_V
        wv = self.w.view(nV, crb, self.ic)
        b, mid = x.shape[0], tuple(x.shape[1:-1])
        raw_w = out.reshape(b, *mid, 1, nV, crb)
        g_w = None if grad is None else grad.reshape(raw_w.shape)
        xs = self._qx(x.reshape(-1, self.ic), a_iv)
        sc = torch.empty(self.eq_n, nV)
        for p0 in range(0, self.eq_n, self.chunk):
            p1 = min(self.eq_n, p0 + self.chunk)
            cur = w_c[p0:p1, :, None, None]                                        # p, n_V, 1, 1
            ws = (torch.clamp(torch.round(wv[None] / cur), -self.wq, self.wq - 1) * cur).reshape(-1, self.ic)
            o = F.linear(xs, ws, None if self.b is None else self.b.repeat(p1 - p0))
            sim = _sim_last(raw_w, o.reshape(b, *mid, p1 - p0, nV, crb), self.metric, g_w)
            sc[p0:p1] = self._reduce(sim, 2)
        return sc

    def score_a(self, x, out, grad, a_c, w_iv):
        """linear.py:497-530 / 609-639: the (eq_n,) table of the activation search, weights quantised with `w_iv` [n_V]."""
        nV, crb = self.n_V, self.oc // self.n_V
        wv = self.w.view(nV, crb, self.ic)
        b, mid = x.shape[0], tuple(x.shape[1:-1])
        x2 = x.reshape(-1, self.ic)
        raw_a = out.reshape(b, *mid, 1, self.oc)
        g_a = None if grad is None else grad.reshape(raw_a.shape)
        ws = (torch.clamp(torch.round(wv / w_iv[:, None, None]), -self.wq, self.wq - 1) * w_iv[:, None, None]).reshape(self.oc, self.ic)
        sc = torch.empty(self.eq_n)
        for p0 in range(0, self.eq_n, self.chunk):
            p1 = min(self.eq_n, p0 + self.chunk)
            cur = a_c[p0:p1].view(1, -1, 1)
            xe = x2[:, None, :]
            if self.postgelu:
                xq = _fq(xe, cur, 0, self.aq - 1) + _fq(xe, self.a_neg, -self.aq, 0)
            else:
                xq = _fq(xe, cur, -self.aq, self.aq - 1)
            o = F.linear(xq, ws, self.b)                                           # rows, p, oc
            sim = _sim_last(raw_a, o.reshape(b, *mid, p1 - p0, self.oc), self.metric, g_a)
            sc[p0:p1] = self._reduce(sim, 1)
        return sc

    def calibration_step2(self, x, out, grad=None):
        """linear.py:536-555."""
        x, out = torch.as_tensor(x, dtype=torch.float32), torch.as_tensor(out, dtype=torch.float32)
        grad = None if grad is None else torch.as_tensor(grad, dtype=torch.float32)
        nV = self.n_V
        w_iv, a_iv, w_c, a_c = self.initial(x)
        for _ in range(self.R):
            sc = self.score_w(x, out, grad, w_c, a_iv)
            self.trace.append(("w0", sc.numpy()))
            w_iv = w_c[sc.argmax(0), torch.arange(nV)]                             # linear.py:493-494
            sc = self.score_a(x, out, grad, a_c, w_iv)
            self.trace.append(("a0", sc.numpy()))
            a_iv = a_c[int(sc.argmax(0))]
        self.w_interval, self.a_interval = w_iv.view(nV, 1, 1, 1).numpy(), a_iv.view(1, 1).numpy()
        return {"w_interval": self.w_interval, "a_interval": self.a_interval}


class TorchMatMul:
    """PTQSLBatchingQuantMatMul / SoSPTQSLBatchingQuantMatMul step 2 (matmul.py:390-644), head-wise, n_V = n_H = 1.

    Pieces for pass-by-pass callers: ``initial``, ``score_split``, ``score_A``, ``score_B``, ``quant_A``."""

    def __init__(self, *, A_bit=8, B_bit=8, metric="hessian", search_round=1, eq_alpha=0.1, eq_beta=2.0, eq_n=100,
                 sos=False, init_layerwise=False, chunk=4, **unused):
        self.Aq, self.Bq = 2 ** (A_bit - 1), 2 ** (B_bit - 1)
        self.metric, self.R, self.eq_n = metric, search_round, eq_n
        self.mult, self.sos, self.init_layerwise, self.chunk = _mult(eq_alpha, eq_beta, eq_n), sos, init_layerwise, chunk
        self.trace = []

    def _sosq(self, A, split):
        """matmul.py:595-598."""
        q1 = self.Aq - 1
        a_int = split / q1
        hi = torch.clamp(torch.round(A.clamp(split, 1) * q1), 0, q1) / q1
        lo = torch.clamp(torch.round(A.clamp(0, split) / a_int), 0, q1) * a_int
        return hi + lo

    def _score(self, out, o, grad):
        """matmul.py:499-503: similarity over the last dim, mean over rows, sum over the batch -> (p, H)."""
        return _sim_last(out[None], o, self.metric, None if grad is None else grad[None]).mean(3).sum(1)

    def initial(self, A, B):
        """matmul.py:419-440 + :567-568: (A_interval [H], B_interval [H], A candidates [eq_n+1, H], B candidates)."""
        H = A.shape[1]
        if self.init_layerwise:
            A_iv = (A.abs().max() / (self.Aq - 0.5)).repeat(H)
            B_iv = (B.abs().max() / (self.Bq - 0.5)).repeat(H)
        else:
            A_iv = A.abs().amax(dim=(0, 2, 3)) / (self.Aq - 0.5)
            B_iv = B.abs().amax(dim=(0, 2, 3)) / (self.Bq - 0.5)
        return A_iv, B_iv, self.mult[:, None] * A_iv[None], self.mult[:, None] * B_iv[None]

    SPLITS = [2.0 ** -i for i in range(20)]

    def score_split(self, A, B, out, grad):
        """matmul.py:600-626: the 20 split scores, against the RAW B."""
        sc = torch.empty(20)
        cands = torch.tensor(self.SPLITS)
        for i in range(20):
            o = self._sosq(A, cands[i]) @ B
            sc[i] = _sim_last(out, o, self.metric, grad).mean(dim=(1, 2)).sum()
        return sc

    def quant_A(self, A, A_iv=None, split=None):
        """The A operand as the B search sees it: matmul.py:595-598 (split) or :124-130 (head-wise interval [H])."""
        if self.sos:
            return self._sosq(A, torch.as_tensor(split, dtype=torch.float32).reshape(()))
        return _fq(A, A_iv.view(1, -1, 1, 1), -self.Aq, self.Aq - 1)

    def score_A(self, A, B, out, grad, A_c, B_iv):
        """matmul.py:483-519: (eq_n, H) table of the A search, B quantised with `B_iv` [H]."""
        H = A.shape[1]
        Bs = _fq(B, B_iv.view(1, H, 1, 1), -self.Bq, self.Bq - 1)
        sc = torch.empty(self.eq_n, H)
        for p0 in range(0, self.eq_n, self.chunk):
            p1 = min(self.eq_n, p0 + self.chunk)
            cur = A_c[p0:p1].view(-1, 1, H, 1, 1)
            sc[p0:p1] = self._score(out, _fq(A[None], cur, -self.Aq, self.Aq - 1) @ Bs[None], grad)
        return sc

    def score_B(self, As, B, out, grad, B_c):
        """matmul.py:524-560: (eq_n, H) table of the B search against the quantised A `As`."""
        H = B.shape[1]
        sc = torch.empty(self.eq_n, H)
        for p0 in range(0, self.eq_n, self.chunk):
            p1 = min(self.eq_n, p0 + self.chunk)
            cur = B_c[p0:p1].view(-1, 1, H, 1, 1)
            sc[p0:p1] = self._score(out, As[None] @ _fq(B[None], cur, -self.Bq, self.Bq - 1), grad)
        return sc

    def calibration_step2(self, A, B, out, grad=None):
        """matmul.py:565-576 / 633-644."""
        A, B, out = (torch.as_tensor(t, dtype=torch.float32) for t in (A, B, out))
        grad = None if grad is None else torch.as_tensor(grad, dtype=torch.float32)
        H = A.shape[1]
        A_iv, B_iv, A_c, B_c = self.initial(A, B)
        split = None
        for _ in range(self.R):
            if self.sos:
                sc = self.score_split(A, B, out, grad)
                self.trace.append(("split", sc.numpy()))
                split = torch.tensor(self.SPLITS)[int(sc.argmax())]
                A_iv = split / (self.Aq - 1)
            else:
                sc = self.score_A(A, B, out, grad, A_c, B_iv)
                self.trace.append(("A", sc.numpy()))
                A_iv = A_c[sc.argmax(0), torch.arange(H)]
            sc = self.score_B(self.quant_A(A, A_iv, split), B, out, grad, B_c)
            self.trace.append(("B", sc.numpy()))
            B_iv = B_c[sc.argmax(0), torch.arange(H)]
        res = {"A_interval": (A_iv.reshape(()) if self.sos else A_iv.view(1, H, 1, 1, 1, 1, 1)).numpy(),
               "B_interval": B_iv.view(1, H, 1, 1, 1, 1, 1).numpy()}
        if self.sos:
            res["split"] = split.numpy()
        return res


class TorchConv:
    """ChannelwiseBatchingQuantConv2d (conv.py:444-614) / BatchingEasyQuantConv2d (conv.py:279-441), a_bit = 32."""

    def __init__(self, weight, bias, *, stride=1, w_bit=8, a_bit=32, metric="hessian", search_round=1, eq_alpha=0.1,
                 eq_beta=2.0, eq_n=100, channelwise=True, init_layerwise=False, chunk=10, **unused):
        assert a_bit >= 32, "the shipped configurations keep the patch embedding's input in fp32 (PTQ4ViT.py:54)"
        self.w = torch.as_tensor(weight, dtype=torch.float32)
        self.b = None if bias is None else torch.as_tensor(bias, dtype=torch.float32)
        self.stride, self.wq, self.metric, self.R, self.eq_n = stride, 2 ** (w_bit - 1), metric, search_round, eq_n
        self.mult, self.channelwise, self.init_layerwise, self.chunk = _mult(eq_alpha, eq_beta, eq_n), channelwise, init_layerwise, chunk
        self.trace = []

    def initial(self):
        """conv.py:482-496 / 312-320 + :594: (w_interval, candidates (eq_n+1, oc | 1, 1, 1, 1))."""
        oc = self.w.shape[0]
        if self.channelwise and not self.init_layerwise:
            w_iv = (self.w.abs().amax(dim=(1, 2, 3)) / (self.wq - 0.5)).view(oc, 1, 1, 1)
        elif self.channelwise:
            w_iv = (self.w.abs().max() / (self.wq - 0.5)).repeat(oc).view(oc, 1, 1, 1)
        else:
            w_iv = self.w.abs().max() / (self.wq - 0.5)
        return w_iv, self.mult.view(-1, 1, 1, 1, 1) * w_iv

    def score_w(self, x, out, grad, w_c):
        """conv.py:526-554 / 365-393: the (eq_n, oc) (channel-wise) or (eq_n,) table of the weight search."""
        oc, b = self.w.shape[0], x.shape[0]
        raw, g = out[:, None], (None if grad is None else grad[:, None])
        sc = torch.empty((self.eq_n, oc) if self.channelwise else (self.eq_n,))
        for p0 in range(0, self.eq_n, self.chunk):
            p1 = min(self.eq_n, p0 + self.chunk)
            p = p1 - p0
            cur = w_c[p0:p1]
            ws = (torch.clamp(torch.round(self.w[None] / cur), -self.wq, self.wq - 1) * cur).reshape(p * oc, *self.w.shape[1:])
            o = F.conv2d(x, ws, None if self.b is None else self.b.repeat(p), self.stride)
            o = o.view(b, p, oc, *o.shape[2:])
            if self.channelwise:                                                # conv.py:498-524, mean over the pixels
                if self.metric == "cosine":
                    s = F.cosine_similarity(raw.flatten(3), o.flatten(3), dim=-1)
                else:
                    s = _elementwise(raw, o, self.metric, g).mean(dim=(3, 4))
            else:                                                               # conv.py:322-351 dim=-3, then the pixels
                s = (F.cosine_similarity(raw, o, dim=2) if self.metric == "cosine"
                     else _elementwise(raw, o, self.metric, g).mean(2)).mean(dim=(2, 3))
            sc[p0:p1] = s.sum(0)
        return sc

    def calibration_step2(self, x, out, grad=None):
        """conv.py:591-607 (the activation search is skipped for a_bit >= 32: :600)."""
        x, out = torch.as_tensor(x, dtype=torch.float32), torch.as_tensor(out, dtype=torch.float32)
        grad = None if grad is None else torch.as_tensor(grad, dtype=torch.float32)
        oc = self.w.shape[0]
        w_iv, w_c = self.initial()
        for _ in range(self.R):
            sc = self.score_w(x, out, grad, w_c)
            self.trace.append(("w", sc.numpy()))
            idx = sc.argmax(0)
            w_iv = w_c[idx, torch.arange(oc)] if self.channelwise else w_c[int(idx)].reshape(())
        self.w_interval = w_iv.numpy()
        return {"w_interval": self.w_interval}
