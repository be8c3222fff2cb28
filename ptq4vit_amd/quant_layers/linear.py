"""Quantised Linear modules -- API mirror of the reference's quant_layers/linear.py.

Hot classes (reference linear.py:349-642): PTQSLBatchingQuantLinear, PostGeluPTQSLBatchingQuantLinear.
Their ``calibration_step2()`` hands weight / bias / raw_input / raw_out / raw_grad to
``p4v_linear_calibrate`` (include/ptq4vit_hip.h): min-max init, candidate grid, the alternating
weight / activation search with the MFMA candidate sweep, argmax + gather all run on the GPU.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import engine
from ._common import POSTGELU_NEG_RANGE, calib_parameters, dispatch, fake_quant, similarity


class MinMaxQuantLinear(nn.Linear):
    """Reference linear.py:6-92."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, mode="raw", w_bit=8, a_bit=8,
                 bias_bit=None, bias_correction=False):
        super().__init__(in_features, out_features, bias)
        assert bias_bit is None, "No support bias bit now"
        self.n_calibration_step = 2
        self.mode = mode
        self.w_bit, self.a_bit, self.bias_bit = w_bit, a_bit, bias_bit
        self.w_interval = None
        self.a_interval = None
        self.raw_input = None
        self.raw_out = None
        self.metric = None
        self.next_nodes = []
        self.w_qmax = 2 ** (w_bit - 1)
        self.a_qmax = 2 ** (a_bit - 1)
        self.bias_correction = bias_correction

    def forward(self, x):
        return dispatch(self, x)

    def raw_forward(self, x):
        return F.linear(x, self.weight, self.bias)

    def quant_weight_bias(self):
        return fake_quant(self.weight, self.w_interval, -self.w_qmax, self.w_qmax - 1), self.bias

    def quant_input(self, x):
        return fake_quant(x, self.a_interval, -self.a_qmax, self.a_qmax - 1)

    def quant_forward(self, x):
        assert self.calibrated is not None, f"You should run calibrate_forward before run quant_forward for {self}"
        w_sim, bias_sim = self.quant_weight_bias()
        return F.linear(self.quant_input(x), w_sim, bias_sim)

    def _bias_correction_quant_forward(self, x):
        if self.bias_correction and self.bias is not None:
            w_sim = self.quant_weight_bias()[0]
            eps = F.linear(self.quant_input(x), w_sim - self.weight.data, None)
            self.bias -= eps.reshape(-1, eps.shape[-1]).mean(0)
            self.bias_correction = False
        return self.quant_forward(x)

    def calibration_step1(self, x):
        out = F.linear(x, self.weight, self.bias)
        self.raw_input, self.raw_out = x.detach(), out.detach()
        return out

    def calibration_step2(self, x):
        self.w_interval = (self.weight.data.abs().max() / (self.w_qmax - 0.5)).detach()
        self.a_interval = (x.abs().max() / (self.a_qmax - 0.5)).detach()
        self.calibrated = True
        return self._bias_correction_quant_forward(x)


class PTQSLQuantLinear(MinMaxQuantLinear):
    """Reference linear.py:94-260 (sub-layerwise: n_V x n_H weight blocks, n_a activation groups)."""

    _postgelu = False

    def __init__(self, in_features: int, out_features: int, bias: bool = True, mode="raw", w_bit=8, a_bit=8,
                 bias_bit=None, bias_correction=False, metric="L2_norm", search_round=1, eq_alpha=0, eq_beta=1,
                 eq_n=100, parallel_eq_n=10, n_H=1, n_V=1, n_a=1, init_layerwise=False):
        super().__init__(in_features, out_features, bias=bias, mode=mode, w_bit=w_bit, a_bit=a_bit,
                         bias_bit=bias_bit, bias_correction=bias_correction)
        self.metric = metric
        self.search_round = search_round
        self.eq_alpha, self.eq_beta, self.eq_n = eq_alpha, eq_beta, eq_n
        self.n_H, self.n_V, self.n_a = n_H, n_V, n_a
        self.crb_rows = out_features // n_V
        self.crb_cols = in_features // n_H
        self.crb_acts = in_features // n_a
        self.parallel_eq_n = parallel_eq_n
        self.init_layerwise = init_layerwise
        self.raw_grad = None

    def quant_weight_bias(self):
        w = self.weight.view(self.n_V, self.crb_rows, self.n_H, self.crb_cols)
        w_sim = fake_quant(w, self.w_interval, -self.w_qmax, self.w_qmax - 1)
        return w_sim.view(self.out_features, self.in_features), self.bias

    def quant_input(self, x):
        xg = x.reshape(*x.shape[:-1], self.n_a, self.crb_acts)
        return fake_quant(xg, self.a_interval, -self.a_qmax, self.a_qmax - 1).reshape(x.shape)

    # ---- int8 inference path --------------------------------------------------
    int8_forward = True   # GPU tensors, no autograd: quant_forward runs as ONE int8 MFMA GEMM (p4v_linear_quant_forward)

    def _positive_a_interval(self):
        return self.a_interval[0] if isinstance(self.a_interval, (list, tuple)) else self.a_interval

    def quant_forward(self, x):
        """Reference linear.py:62-67.  out = Q_a(x) . Q_w(W)^T + bias; on the GPU the product is taken on the integer
        grid indices (exact int32 accumulation) and rescaled by s_a * s_w[block] -- the arithmetic of the candidate
        sweeps; the fake-quant fp32 formulation below is kept for CPU tensors and whenever autograd is recording."""
        assert self.calibrated is not None, f"You should run calibrate_forward before run quant_forward for {self}"
        # what p4v_linear_quant_forward implements (include/ptq4vit_hip.h): 2..8 bit grids, blocks that tile the layer.
        # Inside that envelope the engine's errors propagate -- no silent switch to another arithmetic.
        native = (2 <= self.w_bit <= 8 and 2 <= self.a_bit <= 8 and self.out_features % self.n_V == 0
                  and self.in_features % self.n_H == 0 and self.in_features % self.n_a == 0)
        if (self.int8_forward and native and x.is_cuda
                and not (torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad))):
            return engine.linear_quant_forward(
                weight=self.weight.data, bias=None if self.bias is None else self.bias.data, x=x,
                w_interval=self.w_interval, a_interval=self._positive_a_interval(), w_bit=self.w_bit,
                a_bit=self.a_bit, n_V=self.n_V, n_H=self.n_H, n_a=self.n_a, postgelu=self._postgelu)
        w_sim, bias_sim = self.quant_weight_bias()
        return F.linear(self.quant_input(x), w_sim, bias_sim)

    # ---- the GPU search ---------------------------------------------------------------------
    def _search_job(self, x, raw_out, raw_grad):
        """The prepared p4v_linear_calibrate call (engine.Job): replaces linear.py:536-555 (and :235-260 for the
        non-batching classes).  engine.run_job makes it alone, engine.calibrate_group together with other modules'."""
        if self.metric == "hessian":
            assert raw_grad is not None, "raw_grad is None in _get_similarity!"
        return engine.linear_job(
            weight=self.weight.data, bias=None if self.bias is None else self.bias.data, x=x, out=raw_out,
            grad=raw_grad if self.metric == "hessian" else None, w_bit=self.w_bit, a_bit=self.a_bit,
            metric=self.metric, eq_alpha=self.eq_alpha, eq_beta=self.eq_beta, eq_n=self.eq_n,
            search_round=self.search_round, n_V=self.n_V, n_H=self.n_H, n_a=self.n_a,
            init_layerwise=self.init_layerwise, postgelu=self._postgelu)

    def _search_install(self, job):
        w_iv, a_iv = job.outputs
        dev = self.weight.device if self.weight.is_cuda else w_iv.device
        self.w_interval = w_iv.view(self.n_V, 1, self.n_H, 1).to(dev)
        self._set_a_interval(a_iv.view(self.n_a, 1).to(dev))

    def _search_on_gpu(self, x, raw_out, raw_grad):
        self._search_install(engine.run_job(self._search_job(x, raw_out, raw_grad)))

    def _set_a_interval(self, a_iv):
        self.a_interval = a_iv

    def calibration_step2(self, x):
        self._search_on_gpu(x, self.raw_out, self.raw_grad)
        self.calibrated = True
        out = self._bias_correction_quant_forward(x.to(self.w_interval.device))
        del self.raw_input, self.raw_out, self.raw_grad
        return out


class PostGeluPTQSLQuantLinear(PTQSLQuantLinear):
    """Reference linear.py:262-347: twin uniform quantisation, ``a_interval = [positive (n_a,1), negative scalar]``."""

    _postgelu = True

    def _set_a_interval(self, a_iv):
        self.a_interval = [a_iv, POSTGELU_NEG_RANGE / self.a_qmax]

    def quant_input(self, x):
        xg = x.reshape(*x.shape[:-1], self.n_a, self.crb_acts)
        x_pos = fake_quant(xg, self.a_interval[0], 0, self.a_qmax - 1)
        x_neg = fake_quant(xg, self.a_interval[1], -self.a_qmax, 0)
        return (x_pos + x_neg).reshape(x.shape)


class PTQSLBatchingQuantLinear(PTQSLQuantLinear):
    """Reference linear.py:349-555: calibrates from cached raw_input / raw_out / raw_grad."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.calib_size = None
        self.calib_batch_size = None
        self.calib_need_batching = False

    def _initialize_calib_parameters(self):
        self.calib_size = int(self.raw_input.shape[0])
        numel = 2 * (self.raw_input.numel() + self.raw_out.numel())
        self.calib_batch_size, self.parallel_eq_n, self.calib_need_batching = calib_parameters(numel, self.calib_size)

    def calibration_job(self):
        """calibration_step2 in two halves, so that the calibrator can run the searches of many modules as one
        p4v_calibrate_group call: the prepared engine call ..."""
        self._initialize_calib_parameters()
        return self._search_job(self.raw_input, self.raw_out, self.raw_grad)

    def calibration_install(self, job):
        """... and what follows it (reference linear.py:551-555: intervals set, calibrated, caches dropped)."""
        self._search_install(job)
        self.calibrated = True
        del self.raw_input, self.raw_out, self.raw_grad

    def calibration_step2(self):
        self.calibration_install(engine.run_job(self.calibration_job()))
        return None

    # ---- the reference's per-pass methods, ONE GPU pass each (C ABI p4v_amax_init_linear /
    # p4v_linear_search_w / p4v_linear_search_a).  calibration_step2 above runs the same kernels fused in one call. ----
    def _stepper(self):
        return engine.LinearStepper(
            weight=self.weight.data, bias=None if self.bias is None else self.bias.data, x=self.raw_input,
            out=self.raw_out, grad=self.raw_grad if self.metric == "hessian" else None, w_bit=self.w_bit,
            a_bit=self.a_bit, metric=self.metric, eq_n=self.eq_n, n_V=self.n_V, n_H=self.n_H, n_a=self.n_a,
            init_layerwise=self.init_layerwise, postgelu=self._postgelu)

    def _initialize_intervals(self):
        """Reference linear.py:380-397 (twin: 576-599): min-max intervals from the cached raw_input."""
        w_iv, a_iv = self._stepper().init_intervals()
        self.w_interval = w_iv.view(self.n_V, 1, self.n_H, 1)
        self._set_a_interval(a_iv.view(self.n_a, 1))

    def _search_best_w_interval(self, weight_interval_candidates):
        """Reference linear.py:455-495; candidates (eq_n+1, n_V, 1, n_H, 1)."""
        w_iv, _, _ = self._stepper().search_w(weight_interval_candidates, self.w_interval, self._positive_a_interval())
        self.w_interval = w_iv.view(self.n_V, 1, self.n_H, 1)

    def _search_best_a_interval(self, input_interval_candidates):
        """Reference linear.py:497-533 (twin: 609-642).  The reference builds and indexes this table as
        (n_a, 1, eq_n+1) (linear.py:544, 512); the engine wants candidate-major (eq_n+1, n_a).  Both layouts are
        accepted -- they only coincide for n_a == 1 -- anything else is refused."""
        c = input_interval_candidates
        n_c = self.eq_n + 1
        if c.dim() == 3 and tuple(c.shape) == (self.n_a, 1, n_c):
            c = c.reshape(self.n_a, n_c).t()                  # the reference's layout
        elif c.numel() == n_c * self.n_a and c.shape[0] == n_c:
            c = c.reshape(n_c, self.n_a)                      # candidate-major
        else:
            raise ValueError(f"input_interval_candidates: expected shape ({self.n_a}, 1, {n_c}) (reference layout) or "
                             f"({n_c}, {self.n_a}[, 1]), got {tuple(c.shape)}")
        a_iv, _, _ = self._stepper().search_a(c.contiguous(), self.w_interval, self._positive_a_interval())
        self._set_a_interval(a_iv.view(self.n_a, 1))

    def _get_similarity(self, tensor_raw, tensor_sim, metric=None, raw_grad=None):
        """Reference linear.py:399-424: per-element similarity, mean over the last dim for the difference metrics."""
        metric = metric or self.metric
        sim = similarity(tensor_raw, tensor_sim, metric, raw_grad=raw_grad, dim=-1)
        return sim if metric == "cosine" else sim.mean(dim=-1)


class PostGeluPTQSLBatchingQuantLinear(PTQSLBatchingQuantLinear):
    """Reference linear.py:557-642: ``a_interval`` is the positive range, ``a_neg_interval`` the fixed negative one."""

    _postgelu = True

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.a_neg_interval = POSTGELU_NEG_RANGE / self.a_qmax

    def quant_input(self, x):
        xg = x.reshape(*x.shape[:-1], self.n_a, self.crb_acts)
        x_pos = fake_quant(xg, self.a_interval, 0, self.a_qmax - 1)
        x_neg = fake_quant(xg, self.a_neg_interval, -self.a_qmax, 0)
        return (x_pos + x_neg).reshape(x.shape)


# (utils/quant_calib.py::_groupable: the calibrator may run these modules' searches as one p4v_calibrate_group call -- only while
# calibration_step2 is THIS method, not a user's override)
PTQSLBatchingQuantLinear.calibration_step2._p4v_grouped = True
