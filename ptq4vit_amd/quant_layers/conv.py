"""Quantised Conv2d modules (patch embedding) -- API mirror of the reference's quant_layers/conv.py.

Hot classes (reference conv.py:279-614): ChannelwiseBatchingQuantConv2d (PTQ4ViT config, one interval per
output channel) and BatchingEasyQuantConv2d (BasePTQ config, one interval per layer).
``calibration_step2()`` calls ``p4v_conv_calibrate`` (include/ptq4vit_hip.h).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import engine
from ._common import calib_parameters, dispatch, fake_quant, similarity


class MinMaxQuantConv2d(nn.Conv2d):
    """Reference conv.py:9-89."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=1, padding=0, dilation=1,
                 groups: int = 1, bias: bool = True, padding_mode: str = "zeros", mode="raw", w_bit=8, a_bit=8,
                 bias_bit=None):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, padding_mode)
        assert bias_bit is None, "No support bias bit now"
        self.n_calibration_steps = 2
        self.mode = mode
        self.w_bit, self.a_bit, self.bias_bit = w_bit, a_bit, bias_bit
        self.w_interval = None
        self.a_interval = None
        self.bias_interval = None
        self.raw_input = None
        self.raw_out = None
        self.metric = None
        self.next_nodes = []
        self.w_qmax = 2 ** (w_bit - 1)
        self.a_qmax = 2 ** (a_bit - 1)

    def forward(self, x):
        return dispatch(self, x)

    def _conv(self, x, w, b):
        if x.is_cuda and self._is_patchify(x):
            # non-overlapping patches (the ViT patch embedding): the convolution IS a GEMM over unfolded patches.
            # MIOpen answers this fp32 shape with its naive direct kernel (2-6 ms per call on MI355X, the largest
            # single item of the capture pass); rocBLAS does the same product in ~0.2 ms.
            B, C, H, W = x.shape
            kh, kw = self.kernel_size
            gh, gw = H // kh, W // kw
            cols = x.reshape(B, C, gh, kh, gw, kw).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gw, C * kh * kw)
            out = F.linear(cols, w.reshape(w.shape[0], -1), b)
            return out.reshape(B, gh, gw, -1).permute(0, 3, 1, 2).contiguous()
        return F.conv2d(x, w, b, self.stride, self.padding, self.dilation, self.groups)

    def _is_patchify(self, x):
        pair = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v)
        k, s, p, d = pair(self.kernel_size), pair(self.stride), pair(self.padding), pair(self.dilation)
        return (self.groups == 1 and x.dim() == 4 and k == s and p == (0, 0) and d == (1, 1)
                and x.shape[2] % k[0] == 0 and x.shape[3] % k[1] == 0)

    def raw_forward(self, x):
        return self._conv(x, self.weight, self.bias)

    def quant_weight_bias(self):
        return fake_quant(self.weight, self.w_interval, -self.w_qmax, self.w_qmax - 1), self.bias

    def quant_input(self, x):
        return fake_quant(x, self.a_interval, -self.a_qmax, self.a_qmax - 1)

    def quant_forward(self, x):
        assert self.calibrated is not None, f"You should run calibrate_forward before run quant_forward for {self}"
        w_sim, bias_sim = self.quant_weight_bias()
        return self._conv(self.quant_input(x), w_sim, bias_sim)

    def calibration_step1(self, x):
        out = self.raw_forward(x)
        self.raw_input, self.raw_out = x.detach(), out.detach()
        return out

    def calibration_step2(self, x):
        self.w_interval = (self.weight.data.abs().max() / (self.w_qmax - 0.5)).detach()
        self.a_interval = (x.abs().max() / (self.a_qmax - 0.5)).detach()
        self.calibrated = True
        return self.quant_forward(x)


class PTQSLQuantConv2d(MinMaxQuantConv2d):
    """Reference conv.py:126-277: sub-layerwise weight blocks (n_V x n_H over the (oc, ic*kh*kw) matrix), one activation
    interval, searched by the non-batching `calibration_step2(x)`.

    The search runs on the GPU through the Linear engine: a convolution is the product of the unfolded patches with the
    (oc, ic*kh*kw) weight matrix, fake quantisation is element-wise (so it commutes with the unfolding; zero padding stays
    zero), and the reference's block-by-block greedy sweep with the score taken over the whole output (conv.py:191-220)
    selects what the Linear sweep's per-row-block scores select: a candidate of block (v, h) only changes the output
    channels of row block v, the rest of the sum is the same for all its candidates.  One GPU pass per searched operand and
    round (p4v_linear_search_w / p4v_linear_search_a); intervals and candidate tables as the reference builds them."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=1, padding=0, dilation=1,
                 groups: int = 1, bias: bool = True, padding_mode: str = "zeros", mode="raw", w_bit=8, a_bit=8,
                 bias_bit=None, metric="L2_norm", search_round=1, eq_alpha=0.1, eq_beta=2, eq_n=100,
                 parallel_eq_n=10, n_V=1, n_H=1, init_layerwise=False):
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=padding, dilation=dilation,
                         groups=groups, bias=bias, padding_mode=padding_mode, mode=mode, w_bit=w_bit, a_bit=a_bit,
                         bias_bit=bias_bit)
        self.metric = metric
        self.search_round = search_round
        self.eq_alpha, self.eq_beta, self.eq_n = eq_alpha, eq_beta, eq_n
        self.parallel_eq_n = parallel_eq_n
        self.n_H, self.n_V = n_H, n_V
        self.init_layerwise = init_layerwise
        self.raw_grad = None

    def _blocks(self, w):
        return w.view(self.n_V, self.out_channels // self.n_V, self.n_H, -1)

    def quant_weight_bias(self):
        """Reference conv.py:183-189 (w_interval: n_V, 1, n_H, 1)."""
        w_sim = fake_quant(self._blocks(self.weight), self.w_interval, -self.w_qmax, self.w_qmax - 1)
        return w_sim.view_as(self.weight), self.bias

    def _initialize_intervals(self, x):
        """Reference conv.py:246-251.  The divisors are TENSORS: torch's GPU kernels turn `tensor / python_scalar` into a
        multiplication by the reciprocal, one ulp away from the IEEE division the reference's CPU path (and the engine's
        initialisation of the batching classes) performs."""
        a_div = torch.tensor(self.a_qmax - 0.5, dtype=torch.float32, device=x.device)
        w_div = torch.tensor(self.w_qmax - 0.5, dtype=torch.float32, device=self.weight.device)
        self.a_interval = (x.abs().max() / a_div).detach()
        if self.init_layerwise:
            self.w_interval = (self.weight.abs().max() / w_div).view(1, 1, 1, 1).repeat(self.n_V, 1, self.n_H, 1).detach()
        else:
            self.w_interval = (self._blocks(self.weight.data).abs().amax([1, 3], keepdim=True) / w_div).detach()

    def calibration_step2(self, x):
        """Reference conv.py:253-277."""
        if self.groups != 1 or self.padding_mode != "zeros":
            raise NotImplementedError("ptq4vit_amd: grouped / non-zero-padded convolutions are not implemented on the GPU")
        if self.metric == "cosine" and (self.n_V > 1 or self.n_H > 1):
            # the reference's cosine runs over ALL output channels whatever the block (conv.py:157-160); the Linear
            # engine's is per row block
            raise NotImplementedError("ptq4vit_amd: PTQSLQuantConv2d with the cosine metric needs n_V = n_H = 1 on the GPU")
        if self.metric == "hessian":
            assert self.raw_grad is not None, "raw_grad is None in _get_similarity!"
        dev = self.weight.device
        x = x.to(dev)
        self._initialize_intervals(x)
        oc = self.out_channels
        cols = F.unfold(x, self.kernel_size, dilation=self.dilation, padding=self.padding, stride=self.stride)   # (B, K, L)
        cols = cols.transpose(1, 2).contiguous()                                                                  # (B, L, K)
        to_rows = lambda t: t.to(dev).reshape(t.shape[0], oc, -1).transpose(1, 2).contiguous()                    # (B, L, oc)
        stepper = engine.LinearStepper(
            weight=self.weight.data.reshape(oc, -1), bias=None if self.bias is None else self.bias.data, x=cols,
            out=to_rows(self.raw_out), grad=to_rows(self.raw_grad) if self.metric == "hessian" else None,
            w_bit=self.w_bit, a_bit=self.a_bit, metric=self.metric, eq_n=self.eq_n, n_V=self.n_V, n_H=self.n_H, n_a=1)
        mult = engine.candidate_multipliers(self.eq_alpha, self.eq_beta, self.eq_n, dev)
        weight_interval_candidates = mult.view(-1, 1, 1, 1, 1) * self.w_interval.unsqueeze(0)     # eq_n+1, n_V, 1, n_H, 1
        input_interval_candidates = mult * self.a_interval                                         # eq_n+1
        for _ in range(self.search_round):
            w_iv, _, _ = stepper.search_w(weight_interval_candidates, self.w_interval, self.a_interval.reshape(1))
            self.w_interval = w_iv.view(self.n_V, 1, self.n_H, 1)
            a_iv, _, _ = stepper.search_a(input_interval_candidates, self.w_interval, self.a_interval.reshape(1))
            self.a_interval = a_iv.reshape(())
        self.calibrated = True
        out = self.quant_forward(x)
        del self.raw_input, self.raw_out, self.raw_grad
        return out


class _BatchingConv(PTQSLQuantConv2d):
    _channelwise = True

    def _initialize_calib_parameters(self):
        """Reference conv.py:467-480 (15 GiB budget); attribute parity only."""
        self.calib_size = int(self.raw_input.shape[0])
        numel = 2 * (self.raw_input.numel() + self.raw_out.numel())
        self.calib_batch_size, self.parallel_eq_n, self.calib_need_batching = calib_parameters(numel, self.calib_size, 15)

    def quant_weight_bias(self):
        return fake_quant(self.weight, self.w_interval, -self.w_qmax, self.w_qmax - 1), self.bias

    def quant_forward(self, x):
        assert self.calibrated is not None, f"You should run calibrate_forward before run quant_forward for {self}"
        w_sim, bias_sim = self.quant_weight_bias()
        x_sim = self.quant_input(x) if self.a_bit < 32 else x
        return self._conv(x_sim, w_sim, bias_sim)

    def calibration_job(self):
        """The prepared p4v_conv_calibrate call (engine.Job): replaces conv.py:591-603 / :429-441."""
        if self.groups != 1 or self.padding_mode != "zeros":
            raise NotImplementedError("ptq4vit_amd: grouped / non-zero-padded convolutions are not implemented on the GPU")
        if self.metric == "hessian":
            assert self.raw_grad is not None, "raw_grad is None in _get_similarity!"
        self._initialize_calib_parameters()
        return engine.conv_job(
            weight=self.weight.data, bias=None if self.bias is None else self.bias.data, x=self.raw_input,
            out=self.raw_out, grad=self.raw_grad if self.metric == "hessian" else None, stride=self.stride,
            padding=self.padding, dilation=self.dilation, w_bit=self.w_bit, a_bit=self.a_bit, metric=self.metric,
            eq_alpha=self.eq_alpha, eq_beta=self.eq_beta, eq_n=self.eq_n, search_round=self.search_round,
            channelwise=self._channelwise, init_layerwise=self.init_layerwise)

    def calibration_install(self, job):
        w_iv, a_iv = job.outputs
        self.w_interval = w_iv.view(-1, 1, 1, 1) if self._channelwise else w_iv.reshape(1, 1, 1, 1)
        self.a_interval = a_iv if self.a_bit >= 32 else a_iv.reshape(())
        self.calibrated = True
        del self.raw_input, self.raw_out, self.raw_grad

    def calibration_step2(self):
        self.calibration_install(engine.run_job(self.calibration_job()))

    # ---- the reference's per-pass methods, ONE GPU pass each (C ABI p4v_amax_init_conv,
    # p4v_conv_search_w_channelwise / _layerwise, p4v_conv_search_a).  calibration_step2 runs them fused in one call. ----
    def _stepper(self):
        if self.groups != 1 or self.padding_mode != "zeros":
            raise NotImplementedError("ptq4vit_amd: grouped / non-zero-padded convolutions are not implemented on the GPU")
        return engine.ConvStepper(
            weight=self.weight.data, bias=None if self.bias is None else self.bias.data, x=self.raw_input,
            out=self.raw_out, grad=self.raw_grad if self.metric == "hessian" else None, stride=self.stride,
            padding=self.padding, dilation=self.dilation, w_bit=self.w_bit, a_bit=self.a_bit, metric=self.metric,
            eq_n=self.eq_n, channelwise=self._channelwise, init_layerwise=self.init_layerwise)

    def _w_shape(self, w_iv):
        return w_iv.view(-1, 1, 1, 1) if self._channelwise else w_iv.reshape(1, 1, 1, 1)

    def _initialize_intervals(self):
        """Reference conv.py:482-496 (channel-wise) / 312-320 (layer-wise)."""
        w_iv, a_iv = self._stepper().init_intervals()
        self.w_interval = self._w_shape(w_iv)
        self.a_interval = a_iv if self.a_bit >= 32 else a_iv.reshape(())

    def _search_best_w_interval(self, weight_interval_candidates):
        """Reference conv.py:526-557 (candidates (eq_n+1, oc, 1, 1, 1)) / 365-396 (layer-wise)."""
        w_iv, _, _ = self._stepper().search_w(weight_interval_candidates, self.w_interval, self.a_interval)
        self.w_interval = self._w_shape(w_iv)

    def _search_best_a_interval(self, input_interval_candidates):
        """Reference conv.py:559-589 (channel-wise class with a_bit < 32 only)."""
        a_iv, _, _ = self._stepper().search_a(input_interval_candidates, self.w_interval, self.a_interval)
        self.a_interval = a_iv.reshape(())


class BatchingEasyQuantConv2d(_BatchingConv):
    """Reference conv.py:279-441: layer-wise EasyQuant (BasePTQ config)."""

    _channelwise = False

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.n_V = 1
        self.n_H = 1

    def _get_similarity(self, tensor_raw, tensor_sim, metric=None, dim=-1, raw_grad=None):
        """Reference conv.py:322-351: cosine over `dim`; difference metrics averaged over `dim`."""
        metric = metric or self.metric
        sim = similarity(tensor_raw, tensor_sim, metric, raw_grad=raw_grad, dim=dim)
        return sim if metric == "cosine" else sim.mean(dim=dim)


class ChannelwiseBatchingQuantConv2d(_BatchingConv):
    """Reference conv.py:444-614: one weight interval per output channel (PTQ4ViT config)."""

    _channelwise = True

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.n_V = self.out_channels
        self.n_H = 1

    def _get_similarity(self, tensor_raw, tensor_sim, metric=None, raw_grad=None):
        """Reference conv.py:498-524: tensors (b, p, oc, fh, fw); cosine over the pixels of one (image, channel),
        the difference metrics element-wise (the caller takes the pixel mean)."""
        metric = metric or self.metric
        if metric == "cosine":
            b, p, oc = tensor_sim.shape[:3]
            return similarity(tensor_raw.reshape(b, 1, oc, -1), tensor_sim.reshape(b, p, oc, -1), metric).view(b, p, oc, 1, 1)
        return similarity(tensor_raw, tensor_sim, metric, raw_grad=raw_grad)


class QuantileQuantConv2d(MinMaxQuantConv2d):
    """Reference conv.py:91-124 (quantile instead of max for the min-max init; unused by the shipped configs)."""

    def __init__(self, *args, w_quantile=0.9999, a_quantile=0.9999, **kwargs):
        super().__init__(*args, **kwargs)
        self.w_quantile, self.a_quantile = w_quantile, a_quantile

    def _quantile(self, tensor, quantile):
        if tensor.numel() >= 16777216:
            n = tensor.numel() // 16777216
            return torch.quantile(tensor.view(-1)[: 16777216 * n].view(n, 16777216), quantile, 1).mean()
        return torch.quantile(tensor, quantile)

    def calibration_step2(self, x):
        self.w_interval = (self._quantile(self.weight.data.abs(), self.w_quantile) / (self.w_qmax - 0.5)).detach()
        self.a_interval = (self._quantile(x.abs(), self.a_quantile) / (self.a_qmax - 0.5)).detach()
        self.calibrated = True
        return self.quant_forward(x)


# (utils/quant_calib.py::_groupable, see quant_layers/linear.py)
_BatchingConv.calibration_step2._p4v_grouped = True
