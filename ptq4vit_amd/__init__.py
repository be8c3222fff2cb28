"""ptq4vit_amd -- MI355X-native calibration engine with the PTQ4ViT module API.

Sub-packages mirror the reference's top-level packages: ``quant_layers``, ``utils``, ``configs``.
``install_as_reference_packages()`` aliases them under the reference's names so that scripts written against
hahnyuan/PTQ4ViT (``from utils.quant_calib import HessianQuantCalibrator`` ...) run unchanged.
"""
import importlib
import os
import sys


def configure_runtime(hw_queues=8):
    """Opt-in process setting for launchers (bench.py, tools/): the search runs one module per host thread and HIP stream, and
    the ROCm runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) -- kernels of streams that share a queue
    serialise.  With the pruned passes a module is a chain of small kernels, so more of them in flight pay (8 queues:
    +6 % on ViT-B/224 x 32; the calibrator's default of 4 search streams + 3 capture lanes was tuned with it).  The variable is
    read when HIP initialises, so this must run BEFORE the process touches the GPU; it never overrides the user's value and
    importing the package no longer sets it (a library import must not change the environment of other HIP users in the
    process).  Returns the value in effect, or None (with a warning) when the GPU was already initialised without it."""
    import warnings
    cur = os.environ.get("GPU_MAX_HW_QUEUES")
    if cur is not None:
        return int(cur)
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
        warnings.warn("ptq4vit_amd.configure_runtime(): the GPU is already initialised, GPU_MAX_HW_QUEUES stays at the runtime's "
                      "default (4); call it before the first GPU use or export GPU_MAX_HW_QUEUES=8")
        return None
    os.environ["GPU_MAX_HW_QUEUES"] = str(int(hw_queues))
    return int(hw_queues)


__version__ = "0.1.0"

_SUBMODULES = ("quant_layers", "quant_layers.linear", "quant_layers.matmul", "quant_layers.conv", "utils",
               "utils.net_wrap", "utils.quant_calib", "utils.models", "utils.shard", "utils.integer", "utils.datasets", "configs", "configs.PTQ4ViT",
               "configs.BasePTQ")


def install_as_reference_packages():
    """Make ``import quant_layers`` / ``utils`` / ``configs`` resolve to this package's implementations."""
    for name in _SUBMODULES:
        sys.modules[name] = importlib.import_module("ptq4vit_amd." + name)
    return [n for n in _SUBMODULES]
