"""BasePTQ config: plain uniform quantisation, cosine metric, one search round (reference configs/BasePTQ.py)."""
from ..quant_layers.conv import BatchingEasyQuantConv2d, PTQSLQuantConv2d  # noqa: F401
from ..quant_layers.linear import PostGeluPTQSLBatchingQuantLinear, PTQSLBatchingQuantLinear  # noqa: F401
from ..quant_layers.matmul import PTQSLBatchingQuantMatMul, SoSPTQSLBatchingQuantMatMul  # noqa: F401

bit = 8
conv_fc_name_list = ["qconv", "qlinear_qkv", "qlinear_proj", "qlinear_MLP_1", "qlinear_MLP_2",
                     "qlinear_classifier", "qlinear_reduction"]
matmul_name_list = ["qmatmul_qk", "qmatmul_scorev"]
w_bit = {name: bit for name in conv_fc_name_list}
a_bit = {name: bit for name in conv_fc_name_list}
A_bit = {name: bit for name in matmul_name_list}
B_bit = {name: bit for name in matmul_name_list}

_search = {"metric": "cosine", "eq_alpha": 0.5, "eq_beta": 1.2, "eq_n": 100, "search_round": 1}
ptqsl_conv2d_kwargs = dict(_search, n_V=1, n_H=1)
ptqsl_linear_kwargs = dict(_search, n_V=1, n_H=1, n_a=1)
ptqsl_matmul_kwargs = dict(_search, n_G_A=1, n_V_A=1, n_H_A=1, n_G_B=1, n_V_B=1, n_H_B=1)


def get_module(module_type, *args, **kwargs):
    """type string -> quant module (reference configs/BasePTQ.py:47-62)."""
    if module_type == "qconv":
        kwargs.update(ptqsl_conv2d_kwargs)
        return BatchingEasyQuantConv2d(*args, **kwargs, w_bit=w_bit["qconv"], a_bit=32)
    if "qlinear" in module_type:
        kwargs.update(ptqsl_linear_kwargs)
        if module_type == "qlinear_qkv":
            kwargs["n_V"] *= 3
        return PTQSLBatchingQuantLinear(*args, **kwargs, w_bit=w_bit[module_type], a_bit=a_bit[module_type])
    if "qmatmul" in module_type:
        kwargs.update(ptqsl_matmul_kwargs)
        return PTQSLBatchingQuantMatMul(*args, **kwargs, A_bit=A_bit[module_type], B_bit=B_bit[module_type])
    raise KeyError(module_type)
