"""ptq4vit_amd -- MI355X-native calibration engine with the PTQ4ViT module API.

Sub-packages mirror the reference's top-level packages: ``quant_layers``, ``utils``, ``configs``.
``install_as_reference_packages()`` aliases them under the reference's names so that scripts written against
hahnyuan/PTQ4ViT (``from utils.quant_calib import HessianQuantCalibrator`` ...) run unchanged.
"""
import importlib
import sys

__version__ = "0.1.0"

_SUBMODULES = ("quant_layers", "quant_layers.linear", "quant_layers.matmul", "quant_layers.conv", "utils",
               "utils.net_wrap", "utils.quant_calib", "utils.models", "utils.shard", "utils.integer", "configs", "configs.PTQ4ViT",
               "configs.BasePTQ")


def install_as_reference_packages():
    """Make ``import quant_layers`` / ``utils`` / ``configs`` resolve to this package's implementations."""
    for name in _SUBMODULES:
        sys.modules[name] = importlib.import_module("ptq4vit_amd." + name)
    return [n for n in _SUBMODULES]
