"""Swin definitions (ptq4vit_amd/utils/models.py) -- host-side checks, no GPU."""
import numpy as np
import torch

from ptq4vit_amd.configs import PTQ4ViT
from ptq4vit_amd.quant_layers.linear import PostGeluPTQSLBatchingQuantLinear, PTQSLBatchingQuantLinear
from ptq4vit_amd.quant_layers.matmul import PTQSLBatchingQuantMatMul, SoSPTQSLBatchingQuantMatMul
from ptq4vit_amd.utils import models, net_wrap


def test_window_attention_forward_equals_reference_patch():
    """Same dataflow as the reference's monkey-patched timm forward (utils/models.py:28-56), with and without the
    shift mask: bit-identical on the fixture produced by that function (oracle/gen_golden.py swin)."""
    g = np.load("tests/golden/swin_window_attention.npz", allow_pickle=False)
    blk = models.SwinBlock(24, (14, 14), num_heads=3, window_size=7, shift_size=3)
    att = blk.attn
    att.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")})
    np.testing.assert_array_equal(blk.attn_mask.numpy(), g["mask"])
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        np.testing.assert_array_equal(att(x, blk.attn_mask).numpy(), g["y_mask"])
        np.testing.assert_array_equal(att(x).numpy(), g["y_nomask"])


def test_swin_base_wraps_to_149_modules():
    """24 blocks x (qkv, proj, matmul1, matmul2, fc1, fc2) + 3 reductions + patch embedding + head."""
    with torch.device("meta"):
        net = models.SwinTransformer(img_size=384, window_size=12, embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32))
    names = [n for n, m in net.named_modules()
             if isinstance(m, (torch.nn.Linear, torch.nn.Conv2d, models.MatMul))]
    assert len(names) == 149
    assert sum(n.endswith("downsample.reduction") for n in names) == 3


def test_tiny_swin_wrap_types_and_forward_unchanged():
    net = models.get_net("swin_tiny_patch4_window7_224", device="cpu", img_size=56, embed_dim=24, depths=(2, 2),
                         num_heads=(2, 4), num_classes=10)
    x = torch.randn(2, 3, 56, 56, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        y0 = net(x)
    wrapped = net_wrap.wrap_modules_in_net(net, PTQ4ViT)
    assert len(wrapped) == 2 * 2 * 6 + 1 + 1 + 1
    assert type(wrapped["layers.0.downsample.reduction"]) is PTQSLBatchingQuantLinear
    assert wrapped["layers.0.downsample.reduction"].bias is None
    assert type(wrapped["layers.0.blocks.1.attn.matmul1"]) is PTQSLBatchingQuantMatMul
    assert type(wrapped["layers.1.blocks.0.attn.matmul2"]) is SoSPTQSLBatchingQuantMatMul
    assert type(wrapped["layers.1.blocks.1.mlp.fc2"]) is PostGeluPTQSLBatchingQuantLinear
    assert wrapped["layers.0.blocks.0.attn.qkv"].n_V == 3
    with torch.no_grad():
        assert torch.equal(net(x), y0)           # raw mode: the wrapped net computes what the float net computed
    assert net.layers[0].blocks[1].shift_size == 3 and net.layers[1].blocks[1].shift_size == 0   # 7x7 map: one window
