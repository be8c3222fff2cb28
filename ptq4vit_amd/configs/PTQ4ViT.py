"""PTQ4ViT config: twin uniform quantisation + Hessian-guided metric (reference configs/PTQ4ViT.py).

Module-level attributes are mutated in place by experiment drivers (reference example/test_all.py:53-78),
so they are plain module globals here too.
"""
from ..quant_layers.conv import ChannelwiseBatchingQuantConv2d, PTQSLQuantConv2d  # noqa: F401
from ..quant_layers.linear import PostGeluPTQSLBatchingQuantLinear, PTQSLBatchingQuantLinear
from ..quant_layers.matmul import PTQSLBatchingQuantMatMul, SoSPTQSLBatchingQuantMatMul

no_softmax = False
no_postgelu = False

bit = 8
conv_fc_name_list = ["qconv", "qlinear_qkv", "qlinear_proj", "qlinear_MLP_1", "qlinear_MLP_2",
                     "qlinear_classifier", "qlinear_reduction"]
matmul_name_list = ["qmatmul_qk", "qmatmul_scorev"]
w_bit = {name: bit for name in conv_fc_name_list}
a_bit = {name: bit for name in conv_fc_name_list}
A_bit = {name: bit for name in matmul_name_list}
B_bit = {name: bit for name in matmul_name_list}

_search = {"metric": "hessian", "eq_alpha": 0.01, "eq_beta": 1.2, "eq_n": 100, "search_round": 3}
ptqsl_conv2d_kwargs = dict(_search, n_V=1, n_H=1)
ptqsl_linear_kwargs = dict(_search, n_V=1, n_H=1, n_a=1, bias_correction=True)  # inert on the batching path
ptqsl_matmul_kwargs = dict(_search, n_G_A=1, n_V_A=1, n_H_A=1, n_G_B=1, n_V_B=1, n_H_B=1)


def get_module(module_type, *args, **kwargs):
    """type string -> quant module (reference configs/PTQ4ViT.py:51-80)."""
    if module_type == "qconv":
        kwargs.update(ptqsl_conv2d_kwargs)
        return ChannelwiseBatchingQuantConv2d(*args, **kwargs, w_bit=w_bit["qconv"], a_bit=32)  # input stays fp32
    if "qlinear" in module_type:
        kwargs.update(ptqsl_linear_kwargs)
        cls = PTQSLBatchingQuantLinear
        if module_type == "qlinear_qkv":
            kwargs["n_V"] *= 3  # q, k, v get separate weight intervals
        elif module_type == "qlinear_MLP_2" and not no_postgelu:
            cls = PostGeluPTQSLBatchingQuantLinear
        elif module_type == "qlinear_classifier":
            kwargs["n_V"] = 1
        return cls(*args, **kwargs, w_bit=w_bit[module_type], a_bit=a_bit[module_type])
    if "qmatmul" in module_type:
        kwargs.update(ptqsl_matmul_kwargs)
        cls = PTQSLBatchingQuantMatMul
        if module_type == "qmatmul_scorev" and not no_softmax:
            cls = SoSPTQSLBatchingQuantMatMul
        return cls(*args, **kwargs, A_bit=A_bit[module_type], B_bit=B_bit[module_type])
    raise KeyError(module_type)
