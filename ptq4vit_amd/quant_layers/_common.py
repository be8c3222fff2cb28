"""Shared helpers of the quant module classes."""
import torch

MODES = ("raw", "quant_forward", "calibration_step1", "calibration_step2")
POSTGELU_NEG_RANGE = 0.16997124254703522  # reference quant_layers/linear.py:574


def fake_quant(x, interval, lo, hi):
    """clamp(round(x / interval), lo, hi) * interval -- reference quant_layers/linear.py:167-168."""
    return torch.clamp(torch.round(x / interval), lo, hi) * interval


def dispatch(module, *inputs):
    """Mode dispatch shared by every quant module (reference linear.py:33-44, matmul.py:22-33, conv.py:40-51)."""
    mode = module.mode
    if mode == "raw":
        return module.raw_forward(*inputs)
    if mode == "quant_forward":
        return module.quant_forward(*inputs)
    if mode == "calibration_step1":
        return module.calibration_step1(*inputs)
    if mode == "calibration_step2":
        return module.calibration_step2(*inputs)
    raise NotImplementedError


def calib_parameters(numel_per_calib, calib_size, budget_gib=3):
    """Chunking policy of the reference (linear.py:365-378): kept for attribute parity only --
    the HIP engine tiles the whole calibration set itself and needs no host-side chunking."""
    calib_batch_size = int(calib_size)
    need_batching = False
    while True:
        numel = numel_per_calib / calib_size * calib_batch_size
        parallel_eq_n = int((budget_gib * 1024 * 1024 * 1024 / 4) // numel)
        if parallel_eq_n <= 1:
            need_batching = True
            calib_batch_size //= 2
        else:
            break
    return calib_batch_size, parallel_eq_n, need_batching


def similarity(tensor_raw, tensor_sim, metric, raw_grad=None, dim=-1):
    """Element-wise similarity of the reference's `_get_similarity` (linear.py:399-424, matmul.py:442-481,
    conv.py:322-363): cosine reduces over `dim`; the difference metrics return one value per element and the caller
    takes the means.  Plain torch on whatever device the tensors live on -- a public helper of the module classes;
    the GPU search does NOT go through it (the metric is fused into the sweep epilogue)."""
    if metric == "cosine":
        return torch.nn.functional.cosine_similarity(tensor_raw, tensor_sim, dim=dim)
    diff = tensor_raw - tensor_sim
    if metric == "L1_norm":
        return -diff.abs()
    if metric == "L2_norm":
        return -diff.pow(2)
    if metric == "linear_weighted_L2_norm":
        return -tensor_raw.abs() * diff.pow(2)
    if metric == "square_weighted_L2_norm":
        return -(tensor_raw * diff).pow(2)
    if metric == "hessian":
        assert raw_grad is not None, "raw_grad is None in _get_similarity!"
        return -(raw_grad.reshape_as(tensor_raw) * diff).pow(2)
    raise NotImplementedError(f"metric {metric} not implemented!")
