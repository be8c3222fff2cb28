"""Swap Linear / Conv2d / MatMul modules of a net for quant modules (reference utils/net_wrap.py:39-81)."""
import torch
import torch.nn as nn

from .models import MatMul

MODULE_TYPES = {"qkv": "qlinear_qkv", "proj": "qlinear_proj", "fc1": "qlinear_MLP_1", "fc2": "qlinear_MLP_2",
                "head": "qlinear_classifier", "matmul1": "qmatmul_qk", "matmul2": "qmatmul_scorev",
                "reduction": "qlinear_reduction"}


def _fold_bn(conv_module, bn_module):
    """(weight, bias) of `conv` followed by `bn` in eval mode as one convolution (reference utils/net_wrap.py:8-28):
    y = gamma * (conv(x) - mean) / sqrt(var + eps) + beta  ==  conv'(x) with W' = W * k, b' = (b - mean) * k + beta,
    k = gamma / sqrt(var + eps) per output channel (gamma = 1, beta = 0 when the BatchNorm is not affine)."""
    std = torch.sqrt(bn_module.running_var + bn_module.eps)
    gamma = bn_module.weight if bn_module.affine else torch.ones_like(std)
    beta = bn_module.bias if bn_module.affine else torch.zeros_like(std)
    k = gamma / std
    weight = conv_module.weight.data * k.view(conv_module.out_channels, 1, 1, 1)
    b0 = conv_module.bias if conv_module.bias is not None else torch.zeros_like(std)
    bias = (b0 - bn_module.running_mean) * k + beta
    return weight, bias


def fold_bn_into_conv(conv_module, bn_module):
    """In-place version (reference utils/net_wrap.py:30-36); creates the conv bias if it had none."""
    w, b = _fold_bn(conv_module, bn_module)
    if conv_module.bias is None:
        conv_module.bias = nn.Parameter(b.data)
    else:
        conv_module.bias.data = b.data
    conv_module.weight.data = w.data


def _parent_and_leaf(net, name):
    parent_name, _, leaf = name.rpartition(".")
    try:
        parent = net.get_submodule(parent_name) if parent_name else net
    except AttributeError:
        raise RuntimeError(f"father module {parent_name} not found")
    return parent, leaf


def _make(cfg, m, leaf):
    if isinstance(m, nn.Conv2d):
        new_m = cfg.get_module("qconv", m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding, m.dilation,
                               m.groups, m.bias is not None, m.padding_mode)
    elif isinstance(m, nn.Linear):
        new_m = cfg.get_module(MODULE_TYPES[leaf], m.in_features, m.out_features)  # KeyError for unknown leaf names, like the reference
    elif isinstance(m, MatMul):
        return cfg.get_module(MODULE_TYPES[leaf])
    else:
        return None
    new_m.weight.data = m.weight.data  # shares storage with the float module (reference net_wrap.py:59-60)
    new_m.bias = m.bias
    return new_m


def wrap_modules_in_net(net, cfg):
    """Replace every Conv2d / Linear / MatMul; returns {qualified name: quant module} in named_modules() order."""
    wrapped_modules = {}
    for name, m in list(net.named_modules()):
        if not name:
            continue
        parent, leaf = _parent_and_leaf(net, name)
        new_m = _make(cfg, m, leaf)
        if new_m is not None:
            setattr(parent, leaf, new_m)
            wrapped_modules[name] = new_m
    print("Completed net wrap.")
    return wrapped_modules


def wrap_certain_modules_in_net(net, cfg, layers, modules_to_wrap, wrap_embedding=False):
    """Wrap only the listed leaf names inside transformer blocks `layers` (reference net_wrap.py:83-139)."""
    wrapped_modules = {}
    for name, m in list(net.named_modules()):
        if not name:
            continue
        parent, leaf = _parent_and_leaf(net, name)
        if isinstance(m, nn.Conv2d):
            if not wrap_embedding:
                continue
        else:
            parts = name.split(".")
            in_layer = any(p == "blocks" and i + 1 < len(parts) and parts[i + 1].isdigit() and int(parts[i + 1]) in layers
                           for i, p in enumerate(parts))
            if leaf not in modules_to_wrap or not (in_layer or leaf == "head"):
                continue
        new_m = _make(cfg, m, leaf)
        if new_m is not None:
            setattr(parent, leaf, new_m)
            wrapped_modules[name] = new_m
    print("Completed net wrap.")
    return wrapped_modules
