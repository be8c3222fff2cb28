"""ptq4vit_amd -- MI355X-native calibration engine with the PTQ4ViT module API.

Sub-packages mirror the reference's top-level packages: ``quant_layers``, ``utils``, ``configs``.
``install_as_reference_packages()`` aliases them under the reference's names so that scripts written against
hahnyuan/PTQ4ViT (``from utils.quant_calib import HessianQuantCalibrator`` ...) run unchanged.
"""
import importlib
import os
import sys

# The search runs one module per host thread and HIP stream; the ROCm runtime maps streams onto GPU_MAX_HW_QUEUES hardware
# queues (default 4) and kernels of streams that share a queue serialise.  With the pruned passes a module is a chain of
# small kernels, so more of them in flight pay (8 queues + 8 streams: +6 % on ViT-B/224 x 32).  Only effective when this
# package is imported before the process touches the GPU; never overrides the user's setting.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"

_SUBMODULES = ("quant_layers", "quant_layers.linear", "quant_layers.matmul", "quant_layers.conv", "utils",
               "utils.net_wrap", "utils.quant_calib", "utils.models", "utils.shard", "utils.integer", "configs", "configs.PTQ4ViT",
               "configs.BasePTQ")


def install_as_reference_packages():
    """Make ``import quant_layers`` / ``utils`` / ``configs`` resolve to this package's implementations."""
    for name in _SUBMODULES:
        sys.modules[name] = importlib.import_module("ptq4vit_amd." + name)
    return [n for n in _SUBMODULES]
