"""GPU: which rounding v_cvt_pk_u8_f32 has on this part (the engine probes it once per process, csrc/p4v_api.hip::cvt_bias) and
whether the quantisers built on it reproduce the IEEE-division planes: prints the probe line of the engine and compares
p4v_pack_plane_i8 with quant16_sat8 on / off (tuning 12 = 11) on adversarial inputs (values on and next to rounding breakpoints)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ptq4vit_amd import engine

g = torch.Generator().manual_seed(0)
x = torch.randn(4, 50, 96, generator=g).cuda()
w = (torch.randn(192, 96, generator=g) * 0.05).cuda()
out = torch.nn.functional.linear(x, w)
grad = (torch.randn(out.shape, generator=g) * 1e-3).cuda()
engine.debug_tuning(4, 1)
engine.linear_calibrate(weight=w, bias=None, x=x, out=out, grad=grad, w_bit=8, a_bit=8, metric="hessian", eq_alpha=0.01, eq_beta=1.2,
                        eq_n=100, search_round=1, n_V=1, n_H=1, n_a=1)
torch.cuda.synchronize()
engine.debug_tuning(4, 0)
