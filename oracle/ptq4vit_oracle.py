"""CPU oracle for the PTQ4ViT calibration hot path (numpy, float32).

TEST INFRASTRUCTURE ONLY.  This module is a CPU restatement of the reference's
``calibration_step2`` algorithm for the hot classes named in SURVEY.md §8(a).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline -- the
product path (``ptq4vit_amd``) never imports it and fails loudly when the HIP
extension is missing.

Parity pin: the reference ships no tests or golden vectors, so
this oracle is pinned against fixtures produced by importing the reference
itself in the build container (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``,
checked by ``tests/test_oracle_golden.py``).

Every function cites the reference file:line (relative to the reference repo
root) whose arithmetic it follows.  Arithmetic is float32 throughout, like the
reference; the rounding mode is round-half-to-even (``np.rint`` == ``torch.round``).
Summation order differs from torch's kernels (different BLAS / reduction trees),
so score tables agree to ~1e-6 relative, and the selected candidate index agrees
except at near-ties.
"""
from __future__ import annotations

import itertools
from typing import Dict, List, Optional, Tuple

import numpy as np

F32 = np.float32

ELEMENTWISE_METRICS = (
    "L1_norm",
    "L2_norm",
    "linear_weighted_L2_norm",
    "square_weighted_L2_norm",
    "hessian",
)
POSTGELU_NEG_RANGE = 0.16997124254703522  # quant_layers/linear.py:574


# --------------------------------------------------------------------------- #
# primitives
# --------------------------------------------------------------------------- #
def qmax_of(bit: int) -> int:
    """quant_layers/linear.py:29-30 -- qmax = 2**(bit-1)."""
    return 2 ** (bit - 1)


def candidate_multipliers(eq_alpha: float, eq_beta: float, eq_n: int) -> np.ndarray:
    """quant_layers/linear.py:544 -- python-float grid rounded to float32.

    eq_n+1 entries are built; only indices 0..eq_n-1 are ever searched
    (linear.py:466-467).
    """
    return np.array(
        [eq_alpha + i * (eq_beta - eq_alpha) / eq_n for i in range(eq_n + 1)], dtype=F32
    )


def fake_quant(x: np.ndarray, s, lo: int, hi: int) -> np.ndarray:
    """quant_layers/linear.py:167-168 -- (x / s).round().clamp(lo, hi) * s, float32."""
    q = np.rint(np.asarray(x, dtype=F32) / np.asarray(s, dtype=F32))
    np.clip(q, lo, hi, out=q)
    return (q * np.asarray(s, dtype=F32)).astype(F32, copy=False)


def quant_int(x: np.ndarray, s, lo: int, hi: int) -> np.ndarray:
    """Integer grid index of fake_quant (the value the int8 planes must hold)."""
    q = np.rint(np.asarray(x, dtype=F32) / np.asarray(s, dtype=F32))
    np.clip(q, lo, hi, out=q)
    return q.astype(np.int32)


def twin_planes(x: np.ndarray, s_pos, s_neg, qmax: int) -> Tuple[np.ndarray, np.ndarray]:
    """Integer planes of the post-GELU twin quantiser, quant_layers/linear.py:605-606:
    k_pos = clamp(round(x / s_pos), 0, q-1), k_neg = clamp(round(x / s_neg), -q, 0)."""
    return quant_int(x, s_pos, 0, qmax - 1), quant_int(x, s_neg, -qmax, 0)


def sos_planes(A: np.ndarray, split, qmax: int) -> Tuple[np.ndarray, np.ndarray]:
    """Integer planes of the split-of-softmax quantiser, quant_layers/matmul.py:596-597:
    k_hi = clamp(round(clamp(A, split, 1) * (q-1)), 0, q-1), k_lo = clamp(round(clamp(A, 0, split) / (split/(q-1))), 0, q-1).
    The ranges are NOT disjoint: A >= split gives k_lo = round(split / a_int) (= q-1 up to rounding), A < split gives
    k_hi = round(split (q-1)) (0 when split (q-1) < 0.5) -- SURVEY.md App. A-9."""
    A = np.asarray(A, dtype=F32)
    q1 = F32(qmax - 1)
    split = F32(split)
    a_int = split / q1
    hi = np.clip(np.rint(np.clip(A, split, F32(1)) * q1), 0, qmax - 1).astype(np.int32)
    lo = np.clip(np.rint(np.clip(A, F32(0), split) / a_int), 0, qmax - 1).astype(np.int32)
    return hi, lo


def _cosine(a: np.ndarray, b: np.ndarray, axis: int) -> np.ndarray:
    """torch.nn.functional.cosine_similarity(a, b, dim=axis, eps=1e-8).

    ATen normalises each operand by its clamped 2-norm, then sums the products
    (used at quant_layers/linear.py:407, matmul.py:452, conv.py:330,508).
    """
    a, b = np.broadcast_arrays(a, b)
    na = np.sqrt(np.sum(a * a, axis=axis, keepdims=True, dtype=F32))
    nb = np.sqrt(np.sum(b * b, axis=axis, keepdims=True, dtype=F32))
    na = np.maximum(na, F32(1e-8))
    nb = np.maximum(nb, F32(1e-8))
    return np.sum((a / na) * (b / nb), axis=axis, dtype=F32)


def elementwise_similarity(raw, sim, metric: str, grad=None) -> np.ndarray:
    """quant_layers/linear.py:409-421 -- per-element similarity (no reduction)."""
    if metric == "L1_norm":
        return -np.abs(raw - sim)
    if metric == "L2_norm":
        return -((raw - sim) ** 2)
    if metric == "linear_weighted_L2_norm":
        return -np.abs(raw) * (raw - sim) ** 2
    if metric == "square_weighted_L2_norm":
        return -((raw * (raw - sim)) ** 2)
    if metric == "hessian":
        assert grad is not None, "raw_grad is required by the hessian metric"
        return -((grad * (raw - sim)) ** 2)
    raise NotImplementedError(f"metric {metric} not implemented!")


def similarity_lastdim(raw, sim, metric: str, grad=None) -> np.ndarray:
    """quant_layers/linear.py:399-424 -- similarity reduced over the last axis."""
    if metric == "cosine":
        return _cosine(raw, sim, -1)
    return np.mean(elementwise_similarity(raw, sim, metric, grad), axis=-1, dtype=F32)


def _argmax0(scores: np.ndarray) -> np.ndarray:
    """torch.argmax(dim=0): first index on ties, NaN counts as the maximum."""
    return np.argmax(scores, axis=0)


# --------------------------------------------------------------------------- #
# Linear  (quant_layers/linear.py:349-642)
# --------------------------------------------------------------------------- #
def linear_calib_parameters(raw_input: np.ndarray, raw_out: np.ndarray) -> Tuple[int, int, int]:
    """quant_layers/linear.py:365-378 -- (calib_size, calib_batch_size, parallel_eq_n)."""
    calib_size = int(raw_input.shape[0])
    calib_batch_size = calib_size
    while True:
        numel = 2 * (raw_input.size + raw_out.size) / calib_size * calib_batch_size
        parallel_eq_n = int((3 * 1024 * 1024 * 1024 / 4) // numel)
        if parallel_eq_n <= 1:
            calib_batch_size //= 2
        else:
            break
    return calib_size, calib_batch_size, parallel_eq_n


class LinearOracle:
    """PTQSLBatchingQuantLinear / PostGeluPTQSLBatchingQuantLinear step 2.

    quant_layers/linear.py:349-555 (plain) and :557-642 (post-GELU twin).
    ``x`` has shape (b, *mid, ic); ``out``/``grad`` (b, *mid, oc).
    """

    def __init__(self, weight, bias, *, w_bit=8, a_bit=8, metric="hessian", search_round=1,
                 eq_alpha=0.0, eq_beta=1.0, eq_n=100, n_V=1, n_H=1, n_a=1,
                 init_layerwise=False, postgelu=False, chunk: int = 16, batching: bool = True):
        # batching=False: the non-batching classes PTQSLQuantLinear / PostGeluPTQSLQuantLinear (linear.py:94-347) -- the same
        # search with the scores averaged over batch AND tokens in one mean (linear.py:201,227) instead of mean-then-sum
        self.batching = batching
        self.weight = np.ascontiguousarray(weight, dtype=F32)
        self.bias = None if bias is None else np.ascontiguousarray(bias, dtype=F32)
        self.oc, self.ic = self.weight.shape
        self.w_qmax, self.a_qmax = qmax_of(w_bit), qmax_of(a_bit)
        self.metric, self.search_round = metric, search_round
        self.eq_alpha, self.eq_beta, self.eq_n = eq_alpha, eq_beta, eq_n
        self.n_V, self.n_H, self.n_a = n_V, n_H, n_a
        self.crb_rows, self.crb_cols, self.crb_acts = self.oc // n_V, self.ic // n_H, self.ic // n_a
        self.init_layerwise, self.postgelu = init_layerwise, postgelu
        self.a_neg_interval = POSTGELU_NEG_RANGE / self.a_qmax  # linear.py:574 (python float)
        self.chunk = chunk
        self.w_interval = None  # (n_V,1,n_H,1)
        self.a_interval = None  # (n_a,1)
        self.trace: List[Tuple[str, np.ndarray]] = []

    # ---- fixed-scale quantizers ------------------------------------------------
    def quant_weight(self) -> np.ndarray:
        """linear.py:152-162."""
        wv = self.weight.reshape(self.n_V, self.crb_rows, self.n_H, self.crb_cols)
        return fake_quant(wv, self.w_interval, -self.w_qmax, self.w_qmax - 1).reshape(self.oc, self.ic)

    def quant_input(self, x: np.ndarray) -> np.ndarray:
        """linear.py:164-169; twin: linear.py:601-607."""
        xv = x.reshape(*x.shape[:-1], self.n_a, self.crb_acts)
        if not self.postgelu:
            return fake_quant(xv, self.a_interval, -self.a_qmax, self.a_qmax - 1).reshape(x.shape)
        pos = fake_quant(xv, self.a_interval, 0, self.a_qmax - 1)
        neg = fake_quant(xv, F32(self.a_neg_interval), -self.a_qmax, 0)
        return (pos + neg).reshape(x.shape)

    def quant_forward(self, x: np.ndarray) -> np.ndarray:
        """linear.py:62-67."""
        out = self.quant_input(np.asarray(x, dtype=F32)) @ self.quant_weight().T
        return out + self.bias if self.bias is not None else out

    # ---- step 2 ------------------------------------------------------------------
    def initialize_intervals(self, x: np.ndarray) -> None:
        """linear.py:380-397; twin: linear.py:576-599 (signed max, no abs)."""
        wq, aq = F32(self.w_qmax - 0.5), F32(self.a_qmax - 0.5)
        if self.init_layerwise:
            w0 = np.abs(self.weight).max() / wq
            self.w_interval = np.full((self.n_V, 1, self.n_H, 1), w0, dtype=F32)
            a0 = (x.max() if self.postgelu else np.abs(x).max()) / aq
            self.a_interval = np.full((self.n_a, 1), a0, dtype=F32)
            return
        wv = self.weight.reshape(self.n_V, self.crb_rows, self.n_H, self.crb_cols)
        self.w_interval = (np.abs(wv).max(axis=(1, 3), keepdims=True) / wq).astype(F32)
        xv = x.reshape(-1, self.n_a, self.crb_acts)
        xm = xv.max(axis=(0, 2)) if self.postgelu else np.abs(xv).max(axis=(0, 2))
        self.a_interval = (xm / aq).astype(F32).reshape(self.n_a, 1)

    def _reduce(self, sim: np.ndarray, keep_last: int) -> np.ndarray:
        """linear.py:483-487 / 521-525: mean over middle dims, sum over batch.

        ``sim`` is (b, *mid, [trailing kept dims]); returns the trailing dims.
        """
        if not self.batching:            # linear.py:201 / 227: one mean over every leading dim
            return sim.mean(axis=tuple(range(sim.ndim - keep_last)), dtype=F32)
        mid = tuple(range(1, sim.ndim - keep_last))
        if mid:
            sim = sim.mean(axis=mid, dtype=F32)
        return sim.sum(axis=0, dtype=F32)

    def search_w(self, x, out, grad, w_cands) -> None:
        """linear.py:455-495."""
        b = x.shape[0]
        mid = x.shape[1:-1]
        wv = self.weight.reshape(1, self.n_V, self.crb_rows, self.n_H, self.crb_cols)
        raw = out.reshape(b, *mid, 1, self.n_V, self.crb_rows)
        g = None if grad is None else grad.reshape(raw.shape)
        for h in range(self.n_H):
            scores = np.empty((self.eq_n, self.n_V), dtype=F32)
            x_sim = self.quant_input(x)
            for p0 in range(0, self.eq_n, self.chunk):
                p1 = min(self.eq_n, p0 + self.chunk)
                p = p1 - p0
                cur = np.repeat(self.w_interval[None], p, axis=0)  # p,n_V,1,n_H,1
                cur[:, :, :, h:h + 1, :] = w_cands[p0:p1, :, :, h:h + 1, :]
                w_sim = fake_quant(wv, cur, -self.w_qmax, self.w_qmax - 1).reshape(p * self.oc, self.ic)
                o = x_sim.reshape(-1, self.ic) @ w_sim.T
                if self.bias is not None:
                    o += np.tile(self.bias, p)
                o = o.reshape(b, *mid, p, self.n_V, self.crb_rows)
                sim = similarity_lastdim(raw, o, self.metric, g)  # b,*mid,p,n_V
                scores[p0:p1] = self._reduce(sim, 2)
            self.trace.append((f"w{h}", scores))
            idx = _argmax0(scores)  # n_V
            self.w_interval[:, 0, h, 0] = w_cands[idx, np.arange(self.n_V), 0, h, 0]

    def search_a(self, x, out, grad, a_cands) -> None:
        """linear.py:497-533; twin: linear.py:609-642."""
        b = x.shape[0]
        mid = x.shape[1:-1]
        raw = out.reshape(b, *mid, 1, self.oc)
        g = None if grad is None else grad.reshape(raw.shape)
        xv = x.reshape(*x.shape[:-1], self.n_a, self.crb_acts, 1)
        for a in range(self.n_a):
            scores = np.empty((self.eq_n,), dtype=F32)
            w_sim = self.quant_weight()
            for p0 in range(0, self.eq_n, self.chunk):
                p1 = min(self.eq_n, p0 + self.chunk)
                p = p1 - p0
                cur = np.repeat(self.a_interval[:, :, None], p, axis=2)  # n_a,1,p
                cur[a:a + 1, :, :] = a_cands[a:a + 1, :, p0:p1]
                if not self.postgelu:
                    xs = fake_quant(xv, cur, -self.a_qmax, self.a_qmax - 1)
                else:
                    xs = fake_quant(xv, cur, 0, self.a_qmax - 1) + \
                        fake_quant(xv, F32(self.a_neg_interval), -self.a_qmax, 0)
                # b,*mid,n_a,crb_acts,p -> b,*mid,p,ic
                xs = np.moveaxis(xs, -1, -3).reshape(-1, self.ic)
                o = xs @ w_sim.T
                if self.bias is not None:
                    o += self.bias
                o = o.reshape(b, *mid, p, self.oc)
                sim = similarity_lastdim(raw, o, self.metric, g)  # b,*mid,p
                scores[p0:p1] = self._reduce(sim, 1)
            self.trace.append((f"a{a}", scores))
            idx = int(_argmax0(scores))
            self.a_interval[a, 0] = a_cands[a, 0, idx]

    def calibration_step2(self, raw_input, raw_out, raw_grad=None) -> Dict[str, np.ndarray]:
        """linear.py:536-555."""
        x = np.ascontiguousarray(raw_input, dtype=F32)
        out = np.ascontiguousarray(raw_out, dtype=F32)
        grad = None if raw_grad is None else np.ascontiguousarray(raw_grad, dtype=F32)
        self.initialize_intervals(x)
        mult = candidate_multipliers(self.eq_alpha, self.eq_beta, self.eq_n)
        w_cands = mult.reshape(-1, 1, 1, 1, 1) * self.w_interval[None]  # eq_n+1,n_V,1,n_H,1
        a_cands = mult.reshape(1, 1, -1) * self.a_interval[:, :, None]  # n_a,1,eq_n+1
        self.w_interval = self.w_interval.copy()
        self.a_interval = self.a_interval.copy()
        for _ in range(self.search_round):
            self.search_w(x, out, grad, w_cands)
            self.search_a(x, out, grad, a_cands)
        return {"w_interval": self.w_interval, "a_interval": self.a_interval}


# --------------------------------------------------------------------------- #
# MatMul  (quant_layers/matmul.py:390-644)
# --------------------------------------------------------------------------- #
class MatMulOracle:
    """PTQSLBatchingQuantMatMul / SoSPTQSLBatchingQuantMatMul step 2.

    A: (b,H,d1,d2), B: (b,H,d2,d3), out/grad: (b,H,d1,d3).
    matmul.py:390-576 (head-wise), :578-644 (split-of-softmax on A).
    """

    def __init__(self, *, A_bit=8, B_bit=8, metric="hessian", search_round=1, eq_alpha=0.1,
                 eq_beta=2.0, eq_n=100, n_V_A=1, n_H_A=1, n_V_B=1, n_H_B=1,
                 init_layerwise=False, sos=False, chunk: int = 10, n_G_A=None, n_G_B=None, batching: bool = True):
        # batching=False: PTQSLQuantMatMul / SoSPTQSLQuantMatMul (matmul.py:62-388) -- the group counts n_G are the
        # configured ones (the batching classes force n_G = heads, matmul.py:411-417) and the scores are means over the batch
        self.batching = batching
        self.cfg_n_G_A, self.cfg_n_G_B = (1 if sos else n_G_A), n_G_B      # matmul.py:298: the SoS class forces n_G_A = 1
        self.A_qmax, self.B_qmax = qmax_of(A_bit), qmax_of(B_bit)
        self.metric, self.search_round = metric, search_round
        self.eq_alpha, self.eq_beta, self.eq_n = eq_alpha, eq_beta, eq_n
        self.n_V_A, self.n_H_A, self.n_V_B, self.n_H_B = n_V_A, n_H_A, n_V_B, n_H_B
        if sos:  # matmul.py:586-588
            self.n_V_A = self.n_H_A = 1
        self.init_layerwise, self.sos, self.chunk = init_layerwise, sos, chunk
        self.A_interval = self.B_interval = None
        self.split = None
        self.trace: List[Tuple[str, np.ndarray]] = []

    # ---- geometry: matmul.py:109-122 with n_G = #heads (:411-417) -----------------
    def _padding(self, A, B):
        self.n_G_A, self.n_G_B = A.shape[1], B.shape[1]
        if not self.batching:
            self.n_G_A = self.cfg_n_G_A if self.cfg_n_G_A is not None else 1
            self.n_G_B = self.cfg_n_G_B if self.cfg_n_G_B is not None else 1
        cdiv = lambda a, n: (a + n - 1) // n
        self.crb = {
            "A": (cdiv(A.shape[1], self.n_G_A), cdiv(A.shape[2], self.n_V_A), cdiv(A.shape[3], self.n_H_A)),
            "B": (cdiv(B.shape[1], self.n_G_B), cdiv(B.shape[2], self.n_V_B), cdiv(B.shape[3], self.n_H_B)),
        }
        nA = (self.n_G_A, self.n_V_A, self.n_H_A)
        nB = (self.n_G_B, self.n_V_B, self.n_H_B)
        self.n = {"A": nA, "B": nB}
        self.pad = {
            "A": tuple(c * n - s for c, n, s in zip(self.crb["A"], nA, A.shape[1:])),
            "B": tuple(c * n - s for c, n, s in zip(self.crb["B"], nB, B.shape[1:])),
        }

    def _blocked(self, X, which):
        """matmul.py:494 -- zero-pad and view as (1,b,n_G,crb_g,n_V,crb_r,n_H,crb_c)."""
        pg, pr, pc = self.pad[which]
        Xp = np.pad(X, ((0, 0), (0, pg), (0, pr), (0, pc)))
        nG, nV, nH = self.n[which]
        cg, cr, cc = self.crb[which]
        return Xp.reshape(1, -1, nG, cg, nV, cr, nH, cc)

    def _unblock(self, Xb, shape):
        p = Xb.shape[0]
        b, H, r, c = shape
        Xb = Xb.reshape(p, b, Xb.shape[2] * Xb.shape[3], Xb.shape[4] * Xb.shape[5], Xb.shape[6] * Xb.shape[7])
        return Xb[:, :, :H, :r, :c]

    def _quant_blocked(self, X, which, interval, qmax):
        """matmul.py:124-138."""
        Xq = fake_quant(self._blocked(X, which), interval[None], -qmax, qmax - 1)
        return self._unblock(Xq, X.shape)[0]

    def quant_input_A(self, A):
        if self.sos:
            return self._sos_quant(A, F32(self.split))
        return self._quant_blocked(A, "A", self.A_interval, self.A_qmax)

    def quant_input_B(self, B):
        return self._quant_blocked(B, "B", self.B_interval, self.B_qmax)

    def _sos_quant(self, A, split):
        """matmul.py:595-598 (and :612-615 inside the split search)."""
        q1 = F32(self.A_qmax - 1)
        a_int = F32(split) / q1
        hi = np.clip(np.rint(np.clip(A, split, 1) * q1), 0, self.A_qmax - 1) / q1
        lo = np.clip(np.rint(np.clip(A, 0, split) / a_int), 0, self.A_qmax - 1) * a_int
        return (hi + lo).astype(F32)

    def quant_forward(self, A, B):
        """matmul.py:140-145."""
        return self.quant_input_A(np.asarray(A, F32)) @ self.quant_input_B(np.asarray(B, F32))

    # ---- step 2 -------------------------------------------------------------------
    def initialize_intervals(self, A, B):
        """matmul.py:419-440."""
        self._padding(A, B)
        for which, X, qm in (("A", A, self.A_qmax), ("B", B, self.B_qmax)):
            nG, nV, nH = self.n[which]
            if self.init_layerwise:
                iv = np.full((1, nG, 1, nV, 1, nH, 1), np.abs(X).max() / F32(qm - 0.5), dtype=F32)
            else:
                Xb = self._blocked(X, which)
                iv = (np.abs(Xb).max(axis=(0, 1, 3, 5, 7), keepdims=True)[0] / F32(qm - 0.5)).astype(F32)
            setattr(self, f"{which}_interval", iv)

    def _search_blockwise(self, which, A, B, out, grad, cands):
        """matmul.py:483-522 (A) / :524-563 (B)."""
        qm = self.A_qmax if which == "A" else self.B_qmax
        nG, nV, nH = self.n[which]
        interval = getattr(self, f"{which}_interval")
        X = A if which == "A" else B
        Xb = self._blocked(X, which)
        raw = out[None]
        g = None if grad is None else grad[None]
        for v, h in itertools.product(range(nV), range(nH)):
            other = self.quant_input_B(B)[None] if which == "A" else self.quant_input_A(A)[None]
            scores = np.empty((self.eq_n, X.shape[1]), dtype=F32)
            for p0 in range(0, self.eq_n, self.chunk):
                p1 = min(self.eq_n, p0 + self.chunk)
                cur = np.repeat(interval[None], p1 - p0, axis=0)  # p,1,n_G,1,n_V,1,n_H,1
                cur[:, :, :, :, v:v + 1, :, h:h + 1, :] = cands[p0:p1, :, :, :, v:v + 1, :, h:h + 1, :]
                Xs = self._unblock(fake_quant(Xb, cur, -qm, qm - 1), X.shape)
                o = (Xs @ other) if which == "A" else (other @ Xs)  # p,b,H,d1,d3
                sim = similarity_lastdim(raw, o, self.metric, g)  # p,b,H,d1
                if self.batching:
                    sim = sim.mean(axis=3, dtype=F32).sum(axis=1, dtype=F32)  # p,H
                else:
                    sim = sim.mean(axis=(1, 3), dtype=F32)                    # matmul.py:196 / 231
                scores[p0:p1] = sim
            cg = self.crb[which][0]
            pg = self.pad[which][0]
            grp = np.pad(scores, ((0, 0), (0, pg))).reshape(self.eq_n, nG, cg).mean(-1, dtype=F32)
            self.trace.append((f"{which}{v}{h}", grp))
            idx = _argmax0(grp)  # n_G
            interval[0, :, 0, v, 0, h, 0] = cands[idx, 0, np.arange(nG), 0, v, 0, h, 0]

    def _search_split(self, A, B, out, grad, split_cands):
        """matmul.py:600-631 -- one global split against the UNQUANTISED B."""
        scores = np.empty((len(split_cands),), dtype=F32)
        for i, s in enumerate(split_cands):
            o = self._sos_quant(A, s) @ B
            sim = similarity_lastdim(out, o, self.metric, grad)  # b,H,d1
            scores[i] = sim.mean(axis=(1, 2), dtype=F32).sum(dtype=F32) if self.batching else sim.mean(dtype=F32)  # matmul.py:335
        self.trace.append(("split", scores))
        self.split = F32(split_cands[int(_argmax0(scores))])
        self.A_interval = self.split / F32(self.A_qmax - 1)

    def calibration_step2(self, A, B, raw_out, raw_grad=None):
        """matmul.py:565-576 / :633-644."""
        A = np.ascontiguousarray(A, dtype=F32)
        B = np.ascontiguousarray(B, dtype=F32)
        out = np.ascontiguousarray(raw_out, dtype=F32)
        grad = None if raw_grad is None else np.ascontiguousarray(raw_grad, dtype=F32)
        self.initialize_intervals(A, B)
        mult = candidate_multipliers(self.eq_alpha, self.eq_beta, self.eq_n).reshape(-1, 1, 1, 1, 1, 1, 1, 1)
        B_cands = mult * self.B_interval[None]
        if self.sos:
            split_cands = np.array([2 ** (-i) for i in range(20)], dtype=F32)  # matmul.py:636
        else:
            A_cands = mult * self.A_interval[None]
        for _ in range(self.search_round):
            if self.sos:
                self._search_split(A, B, out, grad, split_cands)
            else:
                self._search_blockwise("A", A, B, out, grad, A_cands)
            self._search_blockwise("B", A, B, out, grad, B_cands)
        res = {"A_interval": self.A_interval, "B_interval": self.B_interval}
        if self.sos:
            res["split"] = self.split
        return res


# --------------------------------------------------------------------------- #
# Conv2d  (quant_layers/conv.py:279-614)
# --------------------------------------------------------------------------- #
def im2col(x: np.ndarray, ksize, stride, padding, dilation) -> Tuple[np.ndarray, int, int]:
    """Unfold (b,ic,H,W) -> (b, fh*fw, ic*kh*kw), column order (ic,kh,kw) == F.conv2d weight order."""
    b, ic, H, W = x.shape
    kh, kw = ksize
    sh, sw = stride
    ph, pw = padding
    dh, dw = dilation
    xp = np.pad(x, ((0, 0), (0, 0), (ph, ph), (pw, pw)))
    fh = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    fw = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    cols = np.empty((b, fh, fw, ic, kh, kw), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            cols[:, :, :, :, i, j] = xp[:, :, i * dh:i * dh + sh * fh:sh, j * dw:j * dw + sw * fw:sw].transpose(0, 2, 3, 1)
    return cols.reshape(b, fh * fw, ic * kh * kw), fh, fw


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class ConvOracle:
    """ChannelwiseBatchingQuantConv2d (conv.py:444-614) / BatchingEasyQuantConv2d (conv.py:279-441).

    groups == 1 only (patch embedding).  x: (b,ic,H,W); out/grad: (b,oc,fh,fw).
    """

    def __init__(self, weight, bias, *, stride=1, padding=0, dilation=1, w_bit=8, a_bit=32,
                 metric="hessian", search_round=1, eq_alpha=0.1, eq_beta=2.0, eq_n=100,
                 channelwise=True, init_layerwise=False, chunk: int = 10):
        self.weight = np.ascontiguousarray(weight, dtype=F32)
        self.bias = None if bias is None else np.ascontiguousarray(bias, dtype=F32)
        self.oc = self.weight.shape[0]
        self.ksize = self.weight.shape[2:]
        self.stride, self.padding, self.dilation = _pair(stride), _pair(padding), _pair(dilation)
        self.w_qmax, self.a_bit = qmax_of(w_bit), a_bit
        self.a_qmax = qmax_of(a_bit)
        self.metric, self.search_round = metric, search_round
        self.eq_alpha, self.eq_beta, self.eq_n = eq_alpha, eq_beta, eq_n
        self.channelwise, self.init_layerwise, self.chunk = channelwise, init_layerwise, chunk
        self.w_interval = self.a_interval = None
        self.trace: List[Tuple[str, np.ndarray]] = []

    def _conv(self, cols, wmat, bias_rep, fh, fw):
        """F.conv2d as GEMM: cols (b,L,K) x wmat (n,K) -> (b,n,fh,fw)."""
        o = cols @ wmat.T
        if bias_rep is not None:
            o = o + bias_rep
        return o.transpose(0, 2, 1).reshape(cols.shape[0], wmat.shape[0], fh, fw)

    def quant_weight(self):
        """conv.py:605-607 / :353-356."""
        return fake_quant(self.weight, self.w_interval, -self.w_qmax, self.w_qmax - 1)

    def quant_input(self, x):
        """conv.py:64-67 (skipped when a_bit >= 32: conv.py:544,612)."""
        if self.a_bit >= 32:
            return x
        return fake_quant(x, self.a_interval, -self.a_qmax, self.a_qmax - 1)

    def quant_forward(self, x):
        x = np.asarray(x, F32)
        cols, fh, fw = im2col(self.quant_input(x), self.ksize, self.stride, self.padding, self.dilation)
        return self._conv(cols, self.quant_weight().reshape(self.oc, -1), self.bias, fh, fw)

    def initialize_intervals(self, x):
        """conv.py:482-496 (channel-wise) / :312-320 (layer-wise)."""
        wq = F32(self.w_qmax - 0.5)
        if self.channelwise and not self.init_layerwise:
            self.w_interval = (np.abs(self.weight).max(axis=(1, 2, 3), keepdims=True) / wq).astype(F32)
        elif self.channelwise:
            self.w_interval = np.full((self.oc, 1, 1, 1), np.abs(self.weight).max() / wq, dtype=F32)
        else:
            self.w_interval = F32(np.abs(self.weight).max() / wq)
        self.a_interval = np.array([np.abs(x).max() / F32(self.a_qmax - 0.5)], dtype=F32)

    def _similarity(self, raw, sim, grad):
        """raw (b,1,oc,fh,fw), sim (b,p,oc,fh,fw) -> channel-wise (b,p,oc) or layer-wise (b,p)."""
        b, p, oc = sim.shape[:3]
        if self.channelwise:  # conv.py:498-524 then mean [3,4] (:549)
            if self.metric == "cosine":
                return _cosine(raw.reshape(b, 1, oc, -1), sim.reshape(b, p, oc, -1), -1)
            return elementwise_similarity(raw, sim, self.metric, grad).mean(axis=(3, 4), dtype=F32)
        # conv.py:322-351 with dim=-3, then mean over [fw,fh] (:388)
        if self.metric == "cosine":
            s = _cosine(raw, sim, 2)
        else:
            s = elementwise_similarity(raw, sim, self.metric, grad).mean(axis=2, dtype=F32)
        return s.mean(axis=(2, 3), dtype=F32)

    def search_w(self, x, out, grad, w_cands):
        """conv.py:526-557 (channel-wise) / :365-396 (layer-wise)."""
        cols, fh, fw = im2col(self.quant_input(x), self.ksize, self.stride, self.padding, self.dilation)
        b = x.shape[0]
        raw = out[:, None]
        g = None if grad is None else grad[:, None]
        scores = np.empty((self.eq_n, self.oc) if self.channelwise else (self.eq_n,), dtype=F32)
        for p0 in range(0, self.eq_n, self.chunk):
            p1 = min(self.eq_n, p0 + self.chunk)
            p = p1 - p0
            w_sim = fake_quant(self.weight[None], w_cands[p0:p1], -self.w_qmax, self.w_qmax - 1)
            bias_rep = None if self.bias is None else np.tile(self.bias, p)
            o = self._conv(cols, w_sim.reshape(p * self.oc, -1), bias_rep, fh, fw).reshape(b, p, self.oc, fh, fw)
            scores[p0:p1] = self._similarity(raw, o, g).sum(axis=0, dtype=F32)
        self.trace.append(("w", scores))
        idx = _argmax0(scores)
        if self.channelwise:
            self.w_interval = w_cands[idx, np.arange(self.oc)].astype(F32)  # oc,1,1,1
        else:
            self.w_interval = F32(w_cands[int(idx)].reshape(()))

    def search_a(self, x, out, grad, a_cands):
        """conv.py:559-589 / :398-427 (only when a_bit < 32)."""
        b = x.shape[0]
        raw = out[:, None]
        g = None if grad is None else grad[:, None]
        wmat = self.quant_weight().reshape(self.oc, -1)
        scores = np.empty((self.eq_n,), dtype=F32)
        for c in range(self.eq_n):
            xs = fake_quant(x, a_cands[c], -self.a_qmax, self.a_qmax - 1)
            cols, fh, fw = im2col(xs, self.ksize, self.stride, self.padding, self.dilation)
            o = self._conv(cols, wmat, self.bias, fh, fw)[:, None]
            s = self._similarity(raw, o, g)
            if self.channelwise:
                s = s.mean(axis=2, dtype=F32)  # conv.py:582 mean over [2,3,4]
            scores[c] = s.sum(axis=0, dtype=F32)[0]
        self.trace.append(("a", scores))
        self.a_interval = np.array([a_cands[int(_argmax0(scores))]], dtype=F32)

    def calibration_step2(self, raw_input, raw_out, raw_grad=None):
        """conv.py:591-603 / :429-441."""
        x = np.ascontiguousarray(raw_input, dtype=F32)
        out = np.ascontiguousarray(raw_out, dtype=F32)
        grad = None if raw_grad is None else np.ascontiguousarray(raw_grad, dtype=F32)
        self.initialize_intervals(x)
        mult = candidate_multipliers(self.eq_alpha, self.eq_beta, self.eq_n)
        if self.channelwise:
            w_cands = mult.reshape(-1, 1, 1, 1, 1) * self.w_interval[None]  # eq_n+1,oc,1,1,1
        else:
            w_cands = mult.reshape(-1, 1, 1, 1, 1) * self.w_interval
        a_cands = mult * self.a_interval[0]
        for _ in range(self.search_round):
            self.search_w(x, out, grad, w_cands)
            if self.a_bit < 32:
                self.search_a(x, out, grad, a_cands)
        return {"w_interval": self.w_interval, "a_interval": self.a_interval}


class PTQSLConvOracle(ConvOracle):
    """PTQSLQuantConv2d (conv.py:126-277): the non-batching sub-layerwise conv search, `calibration_step2(x)`.

    Weight blocks (n_V, oc/n_V, n_H, ic*kh*kw/n_H) searched one (v, h) at a time with the score taken over the WHOLE output
    (conv.py:191-220), one activation interval (conv.py:222-244); `_get_similarity(dim=2)` = mean over oc, then the mean
    over (b, fh, fw).  raw_grad is optional (hessian only).
    """

    def __init__(self, weight, bias, *, n_V=1, n_H=1, **kw):
        kw.setdefault("a_bit", 8)
        super().__init__(weight, bias, channelwise=False, **kw)
        self.n_V, self.n_H = n_V, n_H

    def _wview(self, w):
        return w.reshape(self.n_V, self.oc // self.n_V, self.n_H, -1)

    def quant_weight(self):
        """conv.py:183-189."""
        return fake_quant(self._wview(self.weight), self.w_interval, -self.w_qmax, self.w_qmax - 1).reshape(self.weight.shape)

    def initialize_intervals(self, x):
        """conv.py:246-251."""
        self.a_interval = np.array([np.abs(x).max() / F32(self.a_qmax - 0.5)], dtype=F32)
        wq = F32(self.w_qmax - 0.5)
        if self.init_layerwise:
            self.w_interval = np.full((self.n_V, 1, self.n_H, 1), np.abs(self.weight).max() / wq, dtype=F32)
        else:
            self.w_interval = (np.abs(self._wview(self.weight)).max(axis=(1, 3), keepdims=True) / wq).astype(F32)

    def _sim(self, raw, sim, grad):
        """conv.py:157-181 with dim=2: raw (b,1,oc,fh,fw) / sim (b,p,oc,fh,fw) -> (b,p,fh,fw)."""
        if self.metric == "cosine":
            return _cosine(raw, sim, 2)
        return elementwise_similarity(raw, sim, self.metric, grad).mean(axis=2, dtype=F32)

    def search_w(self, x, out, grad, w_cands):
        cols, fh, fw = im2col(self.quant_input(x), self.ksize, self.stride, self.padding, self.dilation)
        b = x.shape[0]
        raw = out[:, None]
        g = None if grad is None else grad[:, None]
        tmp = self.w_interval.copy()
        for v in range(self.n_V):
            for h in range(self.n_H):
                scores = np.empty((self.eq_n,), dtype=F32)
                for p0 in range(0, self.eq_n, self.chunk):
                    p1 = min(self.eq_n, p0 + self.chunk)
                    p = p1 - p0
                    cur = np.repeat(tmp[None], p, axis=0)
                    cur[:, v:v + 1, :, h:h + 1, :] = w_cands[p0:p1, v:v + 1, :, h:h + 1, :]
                    w_sim = fake_quant(self._wview(self.weight)[None], cur, -self.w_qmax, self.w_qmax - 1)
                    bias_rep = None if self.bias is None else np.tile(self.bias, p)
                    o = self._conv(cols, w_sim.reshape(p * self.oc, -1), bias_rep, fh, fw).reshape(b, p, self.oc, fh, fw)
                    scores[p0:p1] = self._sim(raw, o, g).mean(axis=(0, 2, 3), dtype=F32)
                self.trace.append(("w", scores))
                tmp[v, :, h, :] = w_cands[int(_argmax0(scores)), v, :, h, :]
        self.w_interval = tmp

    def search_a(self, x, out, grad, a_cands):
        raw = out[:, None]
        g = None if grad is None else grad[:, None]
        wmat = self.quant_weight().reshape(self.oc, -1)
        scores = np.empty((self.eq_n,), dtype=F32)
        for c in range(self.eq_n):
            xs = fake_quant(x, a_cands[c], -self.a_qmax, self.a_qmax - 1)
            cols, fh, fw = im2col(xs, self.ksize, self.stride, self.padding, self.dilation)
            o = self._conv(cols, wmat, self.bias, fh, fw)[:, None]
            scores[c] = self._sim(raw, o, g).mean(dtype=F32)
        self.trace.append(("a", scores))
        self.a_interval = np.array([a_cands[int(_argmax0(scores))]], dtype=F32)

    def calibration_step2(self, raw_input, raw_out, raw_grad=None):
        """conv.py:253-277."""
        x = np.ascontiguousarray(raw_input, dtype=F32)
        out = np.ascontiguousarray(raw_out, dtype=F32)
        grad = None if raw_grad is None else np.ascontiguousarray(raw_grad, dtype=F32)
        self.initialize_intervals(x)
        mult = candidate_multipliers(self.eq_alpha, self.eq_beta, self.eq_n)
        w_cands = mult.reshape(-1, 1, 1, 1, 1) * self.w_interval[None]      # eq_n+1, n_V, 1, n_H, 1
        a_cands = mult * self.a_interval[0]
        for _ in range(self.search_round):
            self.search_w(x, out, grad, w_cands)
            self.search_a(x, out, grad, a_cands)
        return {"w_interval": self.w_interval, "a_interval": self.a_interval}
