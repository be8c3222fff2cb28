"""Quantised MatMul modules (q.k^T and attn.v) -- API mirror of the reference's quant_layers/matmul.py.

Hot classes (reference matmul.py:390-644): PTQSLBatchingQuantMatMul (head-wise intervals) and
SoSPTQSLBatchingQuantMatMul (split-of-softmax twin on A).  ``calibration_step2()`` calls
``p4v_matmul_calibrate`` (include/ptq4vit_hip.h).
"""
import torch
import torch.nn.functional as F
from torch import nn

from .. import engine
from ._common import dispatch, fake_quant, similarity


class MinMaxQuantMatMul(nn.Module):
    """Reference matmul.py:8-60."""

    def __init__(self, A_bit=8, B_bit=8, mode="raw"):
        super().__init__()
        self.A_bit, self.B_bit = A_bit, B_bit
        self.A_interval = None
        self.B_interval = None
        self.A_qmax = 2 ** (A_bit - 1)
        self.B_qmax = 2 ** (B_bit - 1)
        self.mode = mode
        self.raw_input = None
        self.raw_out = None

    def forward(self, A, B):
        return dispatch(self, A, B)

    def raw_forward(self, A, B):
        return A @ B

    def quant_input(self, x, interval, qmax):
        return fake_quant(x, interval, -qmax, qmax - 1)

    def quant_forward(self, A, B):
        assert self.calibrated is not None, f"You should run calibrate_forward before run quant_forward for {self}"
        return self.quant_input(A, self.A_interval, self.A_qmax) @ self.quant_input(B, self.B_interval, self.B_qmax)

    def calibration_step1(self, A, B):
        self.raw_input = A.detach(), B.detach()
        out = A @ B
        self.raw_out = out.detach()
        return out

    def calibration_step2(self, A, B):
        self.A_interval = (A.data.abs().max() / (self.A_qmax - 0.5)).detach()
        self.B_interval = (B.data.abs().max() / (self.B_qmax - 0.5)).detach()
        self.calibrated = True
        return self.quant_forward(A, B)


class PTQSLQuantMatMul(MinMaxQuantMatMul):
    """Reference matmul.py:62-282: operands viewed as (n_G, n_V, n_H) blocks with zero padding;
    interval shape (1, n_G, 1, n_V, 1, n_H, 1)."""

    _sos = False

    def __init__(self, A_bit=8, B_bit=8, mode="raw", metric="L2_norm", search_round=1, eq_alpha=0.1, eq_beta=2,
                 eq_n=100, parallel_eq_n=10, n_G_A=1, n_V_A=1, n_H_A=1, n_G_B=1, n_V_B=1, n_H_B=1,
                 init_layerwise=False):
        super().__init__(A_bit=A_bit, B_bit=B_bit, mode=mode)
        self.metric = metric
        self.search_round = search_round
        self.eq_alpha, self.eq_beta, self.eq_n = eq_alpha, eq_beta, eq_n
        self.parallel_eq_n = parallel_eq_n
        self.n_G_A, self.n_V_A, self.n_H_A = n_G_A, n_V_A, n_H_A
        self.n_G_B, self.n_V_B, self.n_H_B = n_G_B, n_V_B, n_H_B
        for s in "AB":
            for what in ("crb_groups", "crb_rows", "crb_cols", "pad_groups", "pad_rows", "pad_cols"):
                setattr(self, f"{what}_{s}", None)
        self.raw_grad = None
        self.init_layerwise = init_layerwise

    def _get_padding_parameters(self, A, B):
        """Reference matmul.py:109-122."""
        for s, X in (("A", A), ("B", B)):
            n = [getattr(self, f"n_{k}_{s}") for k in "GVH"]
            crb = [(X.shape[i + 1] + n[i] - 1) // n[i] for i in range(3)]
            for name, c, ni, dim in zip(("groups", "rows", "cols"), crb, n, X.shape[1:]):
                setattr(self, f"crb_{name}_{s}", c)
                setattr(self, f"pad_{name}_{s}", c * ni - dim)

    def _quant_blocked(self, x, s, interval, qmax):
        """Reference matmul.py:124-138."""
        pg, pr, pc = (getattr(self, f"pad_{k}_{s}") for k in ("groups", "rows", "cols"))
        nG, nV, nH = (getattr(self, f"n_{k}_{s}") for k in "GVH")
        cg, cr, cc = (getattr(self, f"crb_{k}_{s}") for k in ("groups", "rows", "cols"))
        xp = F.pad(x, [0, pc, 0, pr, 0, pg]).view(-1, nG, cg, nV, cr, nH, cc)
        xq = fake_quant(xp, interval, -qmax, qmax - 1).view(-1, nG * cg, nV * cr, nH * cc)
        return xq[:, : xq.shape[1] - pg, : xq.shape[2] - pr, : xq.shape[3] - pc]

    def quant_input_A(self, x):
        return self._quant_blocked(x, "A", self.A_interval, self.A_qmax)

    def quant_input_B(self, x):
        return self._quant_blocked(x, "B", self.B_interval, self.B_qmax)

    int8_forward = True   # GPU tensors, no autograd: quant_forward runs as ONE int8 MFMA GEMM (p4v_matmul_quant_forward)

    def quant_forward(self, A, B):
        """Reference matmul.py:140-145 (SoS operand: matmul.py:595-598).  On the GPU the product is taken on the grid
        indices (head-wise intervals; the split-of-softmax operand as its two range planes) and rescaled in the
        epilogue; the fake-quant fp32 formulation is kept for CPU tensors, autograd and block layouts other than
        head-wise."""
        assert self.calibrated is not None, f"You should run calibrate_forward before run quant_forward for {self}"
        if self.crb_rows_A is None or self.crb_rows_B is None:
            # intervals loaded from elsewhere (shard.exchange_intervals on a rank that did not search this module, or a
            # checkpoint): the block geometry is a function of the operand shapes only (matmul.py:109-122)
            self._get_padding_parameters(A, B)
        # whole heads share a scale (row / column sub-blocks do not): group intervals are handed over head by head
        by_head = (self.n_V_A, self.n_H_A, self.n_V_B, self.n_H_B) == (1, 1, 1, 1) and A.dim() == 4
        if (self.int8_forward and A.is_cuda and by_head and 2 <= self.A_bit <= 8 and 2 <= self.B_bit <= 8
                and not (torch.is_grad_enabled() and (A.requires_grad or B.requires_grad))):
            # inside the envelope of p4v_matmul_quant_forward the engine's errors propagate (no silent fallback)
            H = A.shape[1]
            A_iv = self.A_interval if self._sos else self._per_head(self.A_interval, H, self.crb_groups_A)
            return engine.matmul_quant_forward(A=A, B=B, A_interval=A_iv,
                                               B_interval=self._per_head(self.B_interval, H, self.crb_groups_B),
                                               split=self.split if self._sos else None, A_bit=self.A_bit,
                                               B_bit=self.B_bit, sos=self._sos)
        return self.quant_input_A(A) @ self.quant_input_B(B)

    # ---- the GPU search --------------------------------------------------------------------------
    def _search_job(self, A, B, raw_out, raw_grad):
        """The prepared p4v_matmul_calibrate call (engine.Job): replaces matmul.py:565-576 / :633-644.  The engine searches
        one interval per head (the Batching classes force n_G = heads, matmul.py:411-417) with n_V = n_H = 1."""
        if self.metric == "hessian":
            assert raw_grad is not None, "No raw_grad in PTQSLBatchingQuantMatMul!"
        if (self.n_V_A, self.n_H_A, self.n_V_B, self.n_H_B) != (1, 1, 1, 1):
            raise NotImplementedError("ptq4vit_amd: MatMul row/column sub-blocks (n_V, n_H > 1) are not implemented on the GPU")
        H = A.shape[1]
        self.n_G_A, self.n_G_B = H, H   # head-wise (matmul.py:415-416; also overrides the SoS constructor's n_G_A = 1)
        self._get_padding_parameters(A, B)
        return engine.matmul_job(
            A=A, B=B, out=raw_out, grad=raw_grad if self.metric == "hessian" else None, A_bit=self.A_bit,
            B_bit=self.B_bit, metric=self.metric, eq_alpha=self.eq_alpha, eq_beta=self.eq_beta, eq_n=self.eq_n,
            search_round=self.search_round, sos=self._sos, init_layerwise=self.init_layerwise)

    def _search_install(self, job):
        A_iv, B_iv, split = job.outputs
        H = B_iv.numel()
        self.B_interval = B_iv.view(1, H, 1, 1, 1, 1, 1)
        if self._sos:
            self.split = split.reshape(())
            self.A_interval = A_iv.reshape(())
        else:
            self.A_interval = A_iv.view(1, H, 1, 1, 1, 1, 1)

    def _search_on_gpu(self, A, B, raw_out, raw_grad):
        self._search_install(engine.run_job(self._search_job(A, B, raw_out, raw_grad)))

    @staticmethod
    def _per_head(interval, H, crb_groups):
        """(1, n_G, 1, 1, 1, 1, 1) group intervals -> one value per head (head h belongs to group h // crb_groups)."""
        iv = torch.as_tensor(interval).reshape(-1)
        if iv.numel() == H:
            return iv
        idx = torch.arange(H, device=iv.device) // int(crb_groups)
        return iv[idx]

    def _search_grouped(self, A, B, raw_out, raw_grad):
        """Reference matmul.py:165-282 (SoS: :305-388) with the CONFIGURED group counts (the batching classes force
        n_G = heads): one interval per group of crb_groups consecutive heads, the group's score the mean of its heads'
        scores (matmul.py:199 / 234; zero padding heads only rescale the last group).  Every pass is one GPU sweep with
        per-head candidate tables (p4v_matmul_search_A / _B / p4v_sos_search_split); the per-head score table is folded
        to groups and the first-maximum / NaN-is-maximum selection taken on the device."""
        if (self.n_V_A, self.n_H_A, self.n_V_B, self.n_H_B) != (1, 1, 1, 1):
            raise NotImplementedError("ptq4vit_amd: MatMul row/column sub-blocks (n_V, n_H > 1) are not implemented on the GPU")
        self._get_padding_parameters(A, B)
        st = engine.MatMulStepper(A=A, B=B, out=raw_out, grad=raw_grad if self.metric == "hessian" else None,
                                  A_bit=self.A_bit, B_bit=self.B_bit, metric=self.metric, eq_n=self.eq_n, sos=self._sos,
                                  init_layerwise=self.init_layerwise)
        H, dev = st.H, st.dev
        mult = torch.tensor([self.eq_alpha + i * (self.eq_beta - self.eq_alpha) / self.eq_n for i in range(self.eq_n + 1)],
                            dtype=torch.float32, device=dev)
        A_h, B_h = st.init_intervals()                      # amax / (qmax - 0.5) per head (or layer-wise)
        side = {}
        for s, iv_h, nG, crb in (("A", A_h, self.n_G_A, self.crb_groups_A), ("B", B_h, self.n_G_B, self.crb_groups_B)):
            if iv_h is None:
                continue
            idx = torch.arange(H, device=dev) // int(crb)
            # the division by (qmax - 0.5) is monotone: the group's min-max interval is the largest of its heads'
            iv_g = torch.zeros(nG, dtype=torch.float32, device=dev).index_reduce_(0, idx, iv_h, "amax", include_self=False)
            side[s] = dict(idx=idx, iv=iv_g, cands=mult[:, None] * iv_g[None, :], nG=nG)

        def pick(s, scores):
            d = side[s]
            grp = torch.zeros(self.eq_n, d["nG"], dtype=torch.float32, device=dev).index_add_(1, d["idx"], scores[: self.eq_n])
            best = torch.argmax(grp, dim=0)
            d["iv"] = torch.gather(d["cands"], 0, best[None, :]).reshape(-1)

        split = None
        for _ in range(self.search_round):
            B_cur = side["B"]["iv"][side["B"]["idx"]]
            if self._sos:
                split, A_cur, _, _ = st.search_split()
            else:
                d = side["A"]
                _, scores, _ = st.search_A(d["cands"][:, d["idx"]], d["iv"][d["idx"]], B_cur, want_scores=True)
                pick("A", scores)
                A_cur = d["iv"][d["idx"]]
            d = side["B"]
            _, scores, _ = st.search_B(d["cands"][:, d["idx"]], A_cur, B_cur, split=split, want_scores=True)
            pick("B", scores)
        self.B_interval = side["B"]["iv"].view(1, self.n_G_B, 1, 1, 1, 1, 1)
        if self._sos:
            self.split = split.reshape(())
            self.A_interval = A_cur.reshape(())
        else:
            self.A_interval = side["A"]["iv"].view(1, self.n_G_A, 1, 1, 1, 1, 1)

    def calibration_step2(self, A, B):
        H = A.shape[1]
        if self.n_G_B == H and (self._sos or self.n_G_A == H):
            self._search_on_gpu(A, B, self.raw_out, self.raw_grad)       # head-wise: the fused call
        else:
            self._search_grouped(A, B, self.raw_out, self.raw_grad)
        self.calibrated = True
        del self.raw_input, self.raw_out, self.raw_grad
        dev = self.B_interval.device
        return self.quant_forward(A.to(dev), B.to(dev))


class SoSPTQSLQuantMatMul(PTQSLQuantMatMul):
    """Reference matmul.py:284-388: split-of-softmax twin quantiser on the score matrix A."""

    _sos = True

    def __init__(self, *args, split=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.n_G_A = self.n_V_A = self.n_H_A = 1
        self.A_qmax = 2 ** (self.A_bit - 1)
        self.split = split
        if split is not None:
            self.A_interval = self.split / (self.A_qmax - 1)

    def quant_input_A(self, x):
        q1 = self.A_qmax - 1
        x_high = torch.clamp(torch.round(x.clamp(self.split, 1) * q1), 0, q1) / q1
        x_low = torch.clamp(torch.round(x.clamp(0, self.split) / self.A_interval), 0, q1) * self.A_interval
        return x_high + x_low


class PTQSLBatchingQuantMatMul(PTQSLQuantMatMul):
    """Reference matmul.py:390-576."""

    def _initialize_calib_parameters(self):
        from ._common import calib_parameters
        self.calib_size = int(self.raw_input[0].shape[0])
        numel = self.raw_input[0].numel() + self.raw_input[1].numel() + 2 * self.raw_out.numel()
        self.calib_batch_size, self.parallel_eq_n, self.calib_need_batching = calib_parameters(numel, self.calib_size)

    def _get_padding_parameters(self, A, B):
        """Head-wise quantisation (reference matmul.py:411-417): the group count follows the operands."""
        self.n_G_A = A.shape[1]
        self.n_G_B = B.shape[1]
        super()._get_padding_parameters(A, B)

    def calibration_job(self):
        """The prepared engine call of calibration_step2 (see PTQSLBatchingQuantLinear.calibration_job)."""
        self._initialize_calib_parameters()
        return self._search_job(self.raw_input[0], self.raw_input[1], self.raw_out, self.raw_grad)

    def calibration_install(self, job):
        self._search_install(job)
        self.calibrated = True
        del self.raw_input, self.raw_out, self.raw_grad

    def calibration_step2(self):
        self.calibration_install(engine.run_job(self.calibration_job()))

    # ---- the reference's per-pass methods, ONE GPU pass each (C ABI p4v_amax_init_matmul,
    # p4v_matmul_search_A / _B, p4v_sos_search_split).  calibration_step2 runs the same kernels fused in one call. ----
    def _stepper(self):
        A, B = self.raw_input
        if (self.n_V_A, self.n_H_A, self.n_V_B, self.n_H_B) != (1, 1, 1, 1):
            raise NotImplementedError("ptq4vit_amd: MatMul row/column sub-blocks (n_V, n_H > 1) are not implemented on the GPU")
        self._get_padding_parameters(A, B)
        return engine.MatMulStepper(A=A, B=B, out=self.raw_out, grad=self.raw_grad if self.metric == "hessian" else None,
                                    A_bit=self.A_bit, B_bit=self.B_bit, metric=self.metric, eq_n=self.eq_n,
                                    sos=self._sos, init_layerwise=self.init_layerwise)

    def _initialize_intervals(self):
        """Reference matmul.py:419-440: head-wise min-max intervals (1, heads, 1, 1, 1, 1, 1)."""
        st = self._stepper()
        A_iv, B_iv = st.init_intervals()
        self.B_interval = B_iv.view(1, st.H, 1, 1, 1, 1, 1)
        if A_iv is not None:
            self.A_interval = A_iv.view(1, st.H, 1, 1, 1, 1, 1)

    def _search_best_A_interval(self, A_interval_candidates):
        """Reference matmul.py:483-522; candidates (eq_n+1, 1, heads, 1, 1, 1, 1, 1)."""
        st = self._stepper()
        A_iv, _, _ = st.search_A(A_interval_candidates, self.A_interval, self.B_interval)
        self.A_interval = A_iv.view(1, st.H, 1, 1, 1, 1, 1)

    def _search_best_B_interval(self, B_interval_candidates):
        """Reference matmul.py:524-563 (the split-of-softmax class inherits it with its own quant_input_A)."""
        st = self._stepper()
        B_iv, _, _ = st.search_B(B_interval_candidates, self.A_interval, self.B_interval,
                                 split=self.split if self._sos else None)
        self.B_interval = B_iv.view(1, st.H, 1, 1, 1, 1, 1)

    def _get_similarity(self, tensor_raw, tensor_sim, metric=None, dim=-1, raw_grad=None):
        """Reference matmul.py:442-481: per-element similarity, mean over `dim` for the difference metrics."""
        metric = metric or self.metric
        sim = similarity(tensor_raw, tensor_sim, metric, raw_grad=raw_grad, dim=dim)
        return sim if metric == "cosine" else sim.mean(dim=dim)


class SoSPTQSLBatchingQuantMatMul(PTQSLBatchingQuantMatMul):
    """Reference matmul.py:578-644."""

    _sos = True

    def __init__(self, *args, split=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.n_G_A = self.n_V_A = self.n_H_A = 1
        self.A_qmax = 2 ** (self.A_bit - 1)
        self.split = split
        if split is not None:
            self.A_interval = self.split / (self.A_qmax - 1)

    quant_input_A = SoSPTQSLQuantMatMul.quant_input_A

    def _search_best_A_interval(self, split_candidates=None):
        """Reference matmul.py:600-631: the split search against the unquantised B.  The engine evaluates the
        reference's own grid 2^-i, i = 0..19 (matmul.py:636); any other `split_candidates` is refused."""
        if split_candidates is not None:
            grid = torch.tensor([2.0 ** (-i) for i in range(20)])
            if split_candidates.numel() != 20 or not torch.equal(split_candidates.detach().float().cpu().reshape(-1), grid):
                raise NotImplementedError("ptq4vit_amd: the split search evaluates the grid 2^-i, i = 0..19 (matmul.py:636)")
        split, A_iv, _, _ = self._stepper().search_split()
        self.split = split.reshape(())
        self.A_interval = A_iv.reshape(())

    def calibration_step2(self):
        """Reference matmul.py:633-644 (caches are deleted at the end, :644)."""
        self.calibration_install(engine.run_job(self.calibration_job()))


# (utils/quant_calib.py::_groupable, see quant_layers/linear.py)
PTQSLBatchingQuantMatMul.calibration_step2._p4v_grouped = True
SoSPTQSLBatchingQuantMatMul.calibration_step2._p4v_grouped = True
