"""Integer export of a calibrated model -- API mirror of the reference's utils/integer.py.

The calibrated intervals are the product of the hot path; this module is the data format on its far side: int8
weights, int8 / twin-uint8 activations and the padded sub-layer-wise view of matmul operands.  Same function
names, arguments, dtypes and quirks as the reference (cited per function); tensors may live on the GPU, results
come back on the host exactly like the reference's `.cpu()` calls.

Twin formats, as the reference documents them (integer.py:51-59): post-GELU -- MSB = sign region bit; softmax -- MSB =
large-interval bit.  What the reference COMPUTES, and what is reproduced here bit for bit, is the uint8 sum
`(positive/high index + 128) + (negative/low index)` of both regions (integer.py:63-71, 88-96), uint8 wrap-around
included; pinned by tests/golden/minivit_integer.npz.
"""
import torch
import torch.nn.functional as F

from ..quant_layers.linear import (MinMaxQuantLinear, PostGeluPTQSLBatchingQuantLinear, PostGeluPTQSLQuantLinear)
from ..quant_layers.matmul import (PTQSLBatchingQuantMatMul, PTQSLQuantMatMul, SoSPTQSLBatchingQuantMatMul,
                                   SoSPTQSLQuantMatMul)


def quantize_int_weight(module):
    """int8 weight of a calibrated module (reference integer.py:8-19).  The bias stays fp32.

    `weight / w_interval` broadcasts the (n_V,1,n_H,1) interval tensor against the 2-D weight exactly like the
    reference does, so the result has the reference's (n_V,1,oc,ic)-style shape for sub-layer-wise modules.
    """
    assert hasattr(module, "weight"), f"module {module} does not have weight"
    assert module.w_bit == 8, f"module {module}'s weight is quantized with {module.w_bit} bits"
    w_int = (module.weight / module.w_interval).round_().clamp_(-module.w_qmax, module.w_qmax - 1)
    return w_int.cpu().detach().to(torch.int8)


def dequantize_int_weight(module, w_int):
    """Reference integer.py:21-26: `w_interval * w_int` on the host (same module that produced `w_int`)."""
    return module.w_interval.cpu() * w_int.float()


def quantize_matmul_input(input, interval, qmax, n_G, n_V, n_H, crb_groups, crb_rows, crb_cols):
    """Grid indices of a matmul operand under the sub-layer-wise padding view (reference integer.py:28-43):
    pad (groups, rows, cols) up to n * crb, view as (-1, n_G, crb_g, n_V, crb_r, n_H, crb_c), divide by the
    (1, n_G, 1, n_V, 1, n_H, 1) interval, round, clamp, crop the padding."""
    pad_groups = crb_groups * n_G - input.shape[1]
    pad_rows = crb_rows * n_V - input.shape[2]
    pad_cols = crb_cols * n_H - input.shape[3]
    x = F.pad(input, [0, pad_cols, 0, pad_rows, 0, pad_groups])
    x = x.reshape(-1, n_G, crb_groups, n_V, crb_rows, n_H, crb_cols)
    x = (x / interval).round_().clamp(-qmax, qmax - 1)
    x = x.reshape(-1, n_G * crb_groups, n_V * crb_rows, n_H * crb_cols)
    return x[:, :x.shape[1] - pad_groups, :x.shape[2] - pad_rows, :x.shape[3] - pad_cols]


def quantize_int_activation(module, input):
    """Forward pre-hook storing the integer image of the current inputs in `module.int_input`
    (reference integer.py:46-110).  8-bit only; twin operands use uint8 with the MSB as region bit."""
    if isinstance(module, (PostGeluPTQSLQuantLinear, PostGeluPTQSLBatchingQuantLinear)):
        assert module.a_bit == 8, f"module {module}'s activation is quantized with {module.a_bit} bits"
        x = input[0]
        int_input_pos = (x / module.a_interval).round_().clamp_(0, module.a_qmax - 1)
        int_input_pos = int_input_pos.detach().to(torch.uint8) + 128
        int_input_neg = (x / module.a_neg_interval).round_().clamp_(-module.a_qmax + 1, 0).abs()
        int_input_neg = int_input_neg.detach().to(torch.uint8)
        module.int_input = [(int_input_pos + int_input_neg).cpu()]

    elif isinstance(module, MinMaxQuantLinear):
        assert module.a_bit == 8, f"module {module}'s activation is quantized with {module.a_bit} bits"
        x = input[0]
        int_input = (x / module.a_interval).round_().clamp_(-module.a_qmax, module.a_qmax - 1)
        module.int_input = [int_input.cpu().detach().to(torch.int8)]

    elif isinstance(module, (SoSPTQSLQuantMatMul, SoSPTQSLBatchingQuantMatMul)):
        assert module.A_bit == 8, f"module {module}'s matrix A is quantized with {module.A_bit} bits"
        assert module.B_bit == 8, f"module {module}'s matrix B is quantized with {module.B_bit} bits"
        A, B = input[0], input[1]
        A_high = (A.clamp(module.split, 1) * (module.A_qmax - 1)).round_().clamp_(0, module.A_qmax - 1)
        A_high = A_high.detach().to(torch.uint8) + 128
        A_low = (A.clamp(0, module.split) / module.A_interval).round_().clamp_(0, module.A_qmax - 1)
        A_low = A_low.detach().to(torch.uint8)
        A_int = (A_high + A_low).cpu()
        B_int = quantize_matmul_input(B, module.B_interval, module.B_qmax, module.n_G_B, module.n_V_B, module.n_H_B,
                                      module.crb_groups_B, module.crb_rows_B, module.crb_cols_B)
        module.int_input = [A_int, B_int.cpu().detach().to(torch.int8)]

    elif isinstance(module, (PTQSLQuantMatMul, PTQSLBatchingQuantMatMul)):
        assert module.A_bit == 8, f"module {module}'s matrix A is quantized with {module.A_bit} bits"
        assert module.B_bit == 8, f"module {module}'s matrix B is quantized with {module.B_bit} bits"
        A, B = input[0], input[1]
        A_int = quantize_matmul_input(A, module.A_interval, module.A_qmax, module.n_G_A, module.n_V_A, module.n_H_A,
                                      module.crb_groups_A, module.crb_rows_A, module.crb_cols_A)
        B_int = quantize_matmul_input(B, module.B_interval, module.B_qmax, module.n_G_B, module.n_V_B, module.n_H_B,
                                      module.crb_groups_B, module.crb_rows_B, module.crb_cols_B)
        module.int_input = [A_int.cpu().detach().to(torch.int8), B_int.cpu().detach().to(torch.int8)]


def get_model_int_weight(wrapped_modules):
    """{module name: int8 weight} for every module that has an 8-bit weight (reference integer.py:113-129: modules
    for which `quantize_int_weight` fails -- matmuls, other bit widths -- are skipped silently)."""
    int_weights = {}
    for name, m in wrapped_modules.items():
        try:
            int_weights[name] = quantize_int_weight(m)
        except Exception:
            pass
    return int_weights
