"""Minimal ViT / DeiT definitions that expose the module names the wrapper looks for
(``qkv, proj, fc1, fc2, head, matmul1, matmul2, patch_embed.proj`` -- reference utils/net_wrap.py:42).

The reference fetches timm models and monkey-patches their attention forward so that the two attention
matmuls become nn.Modules (reference utils/models.py:10-26,58-60,79-87).  timm and its pretrained weights are
not available offline, so the architectures are restated here with the same parameter names / shapes as
timm's VisionTransformer; ``get_net(name)`` builds them with seeded random weights (pretrained weights can be
loaded with ``load_state_dict`` when available).
"""
import torch
import torch.nn as nn


class MatMul(nn.Module):
    """A @ B as a module so it can be wrapped (reference utils/models.py:58-60)."""

    def forward(self, A, B):
        return A @ B


class Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias=True):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Identity()
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Identity()
        self.matmul1 = MatMul()
        self.matmul2 = MatMul()

    def forward(self, x):
        """Same dataflow as reference utils/models.py:10-26: q @ k^T through matmul1, attn @ v through matmul2."""
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        attn = self.matmul1(q, k.transpose(-2, -1)) * self.scale
        attn = self.attn_drop(attn.softmax(dim=-1))
        x = self.matmul2(attn, v).transpose(1, 2).reshape(B, N, C)
        return self.proj_drop(self.proj(x))


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch, in_chans, dim):
        super().__init__()
        self.num_patches = (img_size // patch) ** 2
        self.proj = nn.Conv2d(in_chans, dim, kernel_size=patch, stride=patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0):
        super().__init__()
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        self.blocks = nn.Sequential(*[Block(embed_dim, num_heads, mlp_ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head = nn.Linear(embed_dim, num_classes)
        self._init_weights()

    def _init_weights(self):
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Conv2d)):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.patch_embed(x)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1) + self.pos_embed
        x = self.norm(self.blocks(x))
        return self.head(x[:, 0])


_ZOO = {
    # name: (img, patch, dim, depth, heads)
    "vit_tiny_patch16_224": (224, 16, 192, 12, 3),
    "vit_small_patch32_224": (224, 32, 384, 12, 6),
    "vit_small_patch16_224": (224, 16, 384, 12, 6),
    "vit_base_patch16_224": (224, 16, 768, 12, 12),
    "vit_base_patch16_384": (384, 16, 768, 12, 12),
    "deit_tiny_patch16_224": (224, 16, 192, 12, 3),
    "deit_small_patch16_224": (224, 16, 384, 12, 6),
    "deit_base_patch16_224": (224, 16, 768, 12, 12),
    "deit_base_patch16_384": (384, 16, 768, 12, 12),
}


def get_net(name, seed=0, device=None, **overrides):
    """Build a ViT / DeiT by timm name (reference utils/models.py:62-91, without the pretrained download).

    Returns the net in eval mode, on the GPU when one is visible.
    """
    if name not in _ZOO:
        raise KeyError(f"unknown model {name}; known: {sorted(_ZOO)}")
    img, patch, dim, depth, heads = _ZOO[name]
    cfg = dict(img_size=img, patch_size=patch, embed_dim=dim, depth=depth, num_heads=heads)
    cfg.update(overrides)
    g = torch.random.get_rng_state()
    torch.manual_seed(seed)
    net = VisionTransformer(**cfg)
    torch.random.set_rng_state(g)
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    return net.to(device).eval()
