"""GPU: the two drop-in claims of INTEGRATION.md, exercised.

(A) `ptq4vit_amd.install_as_reference_packages()` + a driver written ONLY against the reference's names -- the imports of
    example/test_all.py:1-16, `init_config` with its `reload` (example/test_vit.py:82-90), the `cfg_modifier` that mutates the
    config module in place (example/test_all.py:53-78), `wrap_modules_in_net`, `HessianQuantCalibrator(...).batching_quant_calib()`
    -- run in a fresh interpreter; its intervals must equal, bit for bit, the same calibration written against ptq4vit_amd.*.
(B) the ctypes stub of INTEGRATION.md section B, taken VERBATIM from the file, bound onto a bare nn.Linear subclass that knows
    nothing of this package; its intervals must equal the package's own PTQSLBatchingQuantLinear on the same tensors.
"""
import json
import os
import re
import subprocess
import sys
import textwrap

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
import contextlib, io, json, os, sys
sys.path.insert(0, sys.argv[1])
import torch
import ptq4vit_amd
ptq4vit_amd.install_as_reference_packages()
# ---- from here on: the reference's names only (example/test_all.py:1-16, example/test_vit.py:1-20) ----
from importlib import reload, import_module
from quant_layers.conv import MinMaxQuantConv2d
from quant_layers.linear import MinMaxQuantLinear, PTQSLQuantLinear
from quant_layers.matmul import MinMaxQuantMatMul, PTQSLQuantMatMul
from utils.net_wrap import wrap_certain_modules_in_net
from utils.quant_calib import HessianQuantCalibrator, QuantCalibrator
from utils import net_wrap
from utils.models import get_net


def init_config(config_name):                       # example/test_vit.py:82-90 (the directory walk replaced by the import)
    quant_cfg = import_module(f"configs.{config_name}")
    reload(quant_cfg)
    return quant_cfg


class cfg_modifier():                               # example/test_all.py:48-78
    def __init__(self, **kwargs):
        for name, value in kwargs.items():
            setattr(self, name, value)

    def __call__(self, cfg):
        cfg.bit = self.bit_setting
        cfg.w_bit = {name: self.bit_setting[0] for name in cfg.conv_fc_name_list}
        cfg.a_bit = {name: self.bit_setting[1] for name in cfg.conv_fc_name_list}
        cfg.A_bit = {name: self.bit_setting[1] for name in cfg.matmul_name_list}
        cfg.B_bit = {name: self.bit_setting[1] for name in cfg.matmul_name_list}
        cfg.ptqsl_conv2d_kwargs["n_V"] = self.linear_ptq_setting[0]
        cfg.ptqsl_conv2d_kwargs["n_H"] = self.linear_ptq_setting[1]
        cfg.ptqsl_conv2d_kwargs["metric"] = self.metric
        cfg.ptqsl_conv2d_kwargs["init_layerwise"] = False
        cfg.ptqsl_linear_kwargs["n_V"] = self.linear_ptq_setting[0]
        cfg.ptqsl_linear_kwargs["n_H"] = self.linear_ptq_setting[1]
        cfg.ptqsl_linear_kwargs["n_a"] = self.linear_ptq_setting[2]
        cfg.ptqsl_linear_kwargs["metric"] = self.metric
        cfg.ptqsl_linear_kwargs["init_layerwise"] = False
        cfg.ptqsl_matmul_kwargs["metric"] = self.metric
        cfg.ptqsl_matmul_kwargs["init_layerwise"] = False
        return cfg


def test_all(name, cfg_modifier=lambda x: x, calib_size=32, config_name="PTQ4ViT"):      # example/test_all.py:18-34
    quant_cfg = init_config(config_name)
    quant_cfg = cfg_modifier(quant_cfg)
    net = get_net(name)
    wrapped_modules = net_wrap.wrap_modules_in_net(net, quant_cfg)
    images = torch.randn(calib_size, 3, 224, 224, generator=torch.Generator().manual_seed(0)).cuda()
    calib_loader = [(images, torch.zeros(calib_size, dtype=torch.long))]
    quant_calibrator = HessianQuantCalibrator(net, wrapped_modules, calib_loader, sequential=False, batch_size=4)
    quant_calibrator.batching_quant_calib()
    return net, wrapped_modules


out = {}
with contextlib.redirect_stdout(io.StringIO()):
    for bits in ((8, 8), (6, 6)):
        # a config mutated by the previous experiment must come back fresh from reload (the reference relies on it)
        net, wrapped = test_all("deit_tiny_patch16_224", cfg_modifier(linear_ptq_setting=(1, 1, 1), metric="hessian", bit_setting=bits), 8)
        assert all(isinstance(m, (MinMaxQuantConv2d, MinMaxQuantLinear, MinMaxQuantMatMul)) for m in wrapped.values())
        assert wrapped["blocks.0.attn.qkv"].w_bit == bits[0] and wrapped["blocks.0.attn.matmul1"].A_bit == bits[1]
        iv = {}
        for n, m in wrapped.items():
            for a in ("w_interval", "a_interval", "A_interval", "B_interval", "split"):
                v = getattr(m, a, None)
                if v is not None and not isinstance(v, (list, tuple)):
                    iv[f"{n}.{a}"] = torch.as_tensor(v).detach().float().cpu().reshape(-1).tolist()
        out[str(bits[0])] = iv
fresh = init_config("PTQ4ViT")
assert fresh.bit == 8 and all(v == 8 for v in fresh.w_bit.values()), "reload did not restore the config module"
json.dump(out, open(sys.argv[2], "w"))
'''


def _direct(bits, calib):
    """The same calibration against ptq4vit_amd.* directly."""
    import contextlib
    import io
    from ptq4vit_amd.configs import PTQ4ViT
    from ptq4vit_amd.utils import models, net_wrap
    from ptq4vit_amd.utils.quant_calib import HessianQuantCalibrator
    saved = (PTQ4ViT.bit, dict(PTQ4ViT.w_bit), dict(PTQ4ViT.a_bit), dict(PTQ4ViT.A_bit), dict(PTQ4ViT.B_bit))
    PTQ4ViT.bit = bits
    for tab in (PTQ4ViT.w_bit, PTQ4ViT.a_bit, PTQ4ViT.A_bit, PTQ4ViT.B_bit):
        for k in tab:
            tab[k] = bits
    try:
        net = models.get_net("deit_tiny_patch16_224")
        with contextlib.redirect_stdout(io.StringIO()):
            wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
    finally:
        PTQ4ViT.bit = saved[0]
        for tab, old in zip((PTQ4ViT.w_bit, PTQ4ViT.a_bit, PTQ4ViT.A_bit, PTQ4ViT.B_bit), saved[1:]):
            tab.clear()
            tab.update(old)
    images = torch.randn(calib, 3, 224, 224, generator=torch.Generator().manual_seed(0)).cuda()
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        HessianQuantCalibrator(net, wrapped, [(images, None)], sequential=False, batch_size=4).batching_quant_calib()
    iv = {}
    for n, m in wrapped.items():
        for a in ("w_interval", "a_interval", "A_interval", "B_interval", "split"):
            v = getattr(m, a, None)
            if v is not None and not isinstance(v, (list, tuple)):
                iv[f"{n}.{a}"] = torch.as_tensor(v).detach().float().cpu().reshape(-1).tolist()
    return iv


def test_a_driver_written_against_the_reference_names_runs_on_the_aliased_packages(tmp_path):
    drv, res = tmp_path / "driver.py", tmp_path / "intervals.json"
    drv.write_text(DRIVER)
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    p = subprocess.run([sys.executable, str(drv), ROOT, str(res)], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    got = json.load(open(res))
    n = 0
    for bits in (8, 6):
        want = _direct(bits, 8)
        assert set(got[str(bits)]) == set(want)
        for k, v in want.items():
            assert got[str(bits)][k] == v, f"W{bits}A{bits} {k}: driver {got[str(bits)][k][:3]} vs direct {v[:3]}"
            n += len(v)
    print(f"[drop-in] reference-named driver (reload + cfg_modifier + wrap + calibrate, W8A8 then W6A6): {n} interval scalars bit-identical "
          "to the same calibration written against ptq4vit_amd.*")


def _integration_stub():
    """The python block of INTEGRATION.md section B that defines `calibration_step2`, verbatim."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## B."):]
    block = re.search(r"```python\n(# quant_layers/linear.py.*?)```", sec, re.S).group(1)
    return block


def test_the_ctypes_stub_of_integration_md_binds_onto_a_bare_linear_subclass():
    from ptq4vit_amd import _lib
    from ptq4vit_amd.quant_layers.linear import PTQSLBatchingQuantLinear
    _lib.load()                                                     # (torch first, then the library: one HIP runtime per process)
    src = _integration_stub().replace('C.CDLL("libptq4vit_hip.so")', f'C.CDLL({_lib.LIB_PATH!r})')
    assert "p4v_linear_calibrate" in src and "def calibration_step2(self):" in src

    class PostGeluPTQSLBatchingQuantLinear(torch.nn.Linear):        # (the stub names the reference's twin class in an isinstance)
        pass

    ns = {"PostGeluPTQSLBatchingQuantLinear": PostGeluPTQSLBatchingQuantLinear}
    exec(compile(src, "INTEGRATION.md#B", "exec"), ns)

    class BareLinear(torch.nn.Linear):
        """What the reference's class is to the stub: an nn.Linear with the search hyper-parameters and the cached tensors."""
        calibration_step2 = ns["calibration_step2"]

    g = torch.Generator().manual_seed(21)
    K, N, b, T = 192, 384, 8, 197
    hp = dict(w_bit=8, a_bit=8, metric="hessian", search_round=3, eq_alpha=0.01, eq_beta=1.2, eq_n=100, n_V=3, n_H=1, n_a=1)
    ours = PTQSLBatchingQuantLinear(K, N, mode="raw", bias_bit=None, parallel_eq_n=10, init_layerwise=False, **hp).cuda()
    with torch.no_grad():
        ours.weight.copy_(torch.randn(N, K, generator=g) * 0.05)
        ours.bias.copy_(torch.randn(N, generator=g) * 0.1)
    bare = BareLinear(K, N).cuda()
    bare.load_state_dict(ours.state_dict())
    for k, v in hp.items():
        setattr(bare, k, v)
    bare.init_layerwise = False
    x = torch.randn(b, T, K, generator=g).cuda()
    out = torch.nn.functional.linear(x, ours.weight, ours.bias).detach()
    grad = (torch.randn(out.shape, generator=g) * 1e-3).cuda()
    for m in (ours, bare):
        m.raw_input, m.raw_out, m.raw_grad = x.clone(), out.clone(), grad.clone()
    ours.calibration_step2()
    bare.calibration_step2()
    torch.cuda.synchronize()
    assert bare.calibrated and not hasattr(bare, "raw_input")
    assert torch.equal(bare.w_interval, ours.w_interval) and torch.equal(bare.a_interval, ours.a_interval)
    print(f"[drop-in] INTEGRATION.md section B stub on a bare nn.Linear subclass: w_interval {bare.w_interval.flatten().tolist()}, "
          f"a_interval {bare.a_interval.flatten().tolist()} == the package's own class")
