"""Minimal ViT / DeiT / Swin definitions that expose the module names the wrapper looks for
(``qkv, proj, fc1, fc2, head, reduction, matmul1, matmul2, patch_embed.proj`` -- reference utils/net_wrap.py:42).

The reference fetches timm models and monkey-patches their attention forward so that the two attention
matmuls become nn.Modules (reference utils/models.py:10-26,58-60,79-87).  timm and its pretrained weights are
not available offline, so the architectures are restated here with the same parameter names / shapes as
timm's VisionTransformer; ``get_net(name)`` builds them with seeded random weights (pretrained weights can be
loaded with ``load_state_dict`` when available).
"""
import os

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
        if os.environ.get("P4V_QKV_CONTIGUOUS", "1") == "1":
            # ONE copy into [3][B][H][N][D] instead of the three that torch.matmul makes of the strided q / k / v views (and again of
            # the saved views in its backward): same values, ~5 fewer kernels per block and pass of the capture
            qkv = qkv.contiguous()
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


# ---- Swin Transformer (timm swin_transformer.py names: layers.{i}.blocks.{j}.attn.qkv, ...downsample.reduction) ----
def window_partition(x, ws):
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def window_reverse(windows, ws, H, W):
    B = int(windows.shape[0] / (H * W / ws / ws))
    x = windows.view(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


class WindowAttention(nn.Module):
    """Window attention with relative position bias; forward = reference utils/models.py:28-56
    (`window_attention_forward`): q is scaled BEFORE matmul1, the bias (and the shift mask) are added to its
    output, softmax, matmul2."""

    def __init__(self, dim, window_size, num_heads, qkv_bias=True):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.relative_position_bias_table = nn.Parameter(
            torch.zeros((2 * window_size[0] - 1) * (2 * window_size[1] - 1), num_heads))
        coords = torch.stack(torch.meshgrid(torch.arange(window_size[0]), torch.arange(window_size[1]), indexing="ij"))
        flat = torch.flatten(coords, 1)
        rel = (flat[:, :, None] - flat[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += window_size[0] - 1
        rel[:, :, 1] += window_size[1] - 1
        rel[:, :, 0] *= 2 * window_size[1] - 1
        self.register_buffer("relative_position_index", rel.sum(-1))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Identity()
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Identity()
        self.softmax = nn.Softmax(dim=-1)
        self.matmul1 = MatMul()
        self.matmul2 = MatMul()

    def forward(self, x, mask=None):
        B_, N, C = x.shape
        qkv = self.qkv(x).reshape(B_, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        q = q * self.scale
        attn = self.matmul1(q, k.transpose(-2, -1))
        n = self.window_size[0] * self.window_size[1]
        bias = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(n, n, -1)
        attn = attn + bias.permute(2, 0, 1).contiguous().unsqueeze(0)
        if mask is not None:
            nW = mask.shape[0]
            attn = attn.view(B_ // nW, nW, self.num_heads, N, N) + mask.unsqueeze(1).unsqueeze(0)
            attn = attn.view(-1, self.num_heads, N, N)
        attn = self.attn_drop(self.softmax(attn))
        x = self.matmul2(attn, v).transpose(1, 2).reshape(B_, N, C)
        return self.proj_drop(self.proj(x))


class SwinBlock(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size, shift_size, mlp_ratio=4.0):
        super().__init__()
        self.input_resolution = input_resolution
        if min(input_resolution) <= window_size:      # window covers the whole map: no shift, one window
            shift_size, window_size = 0, min(input_resolution)
        self.window_size, self.shift_size = window_size, shift_size
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, (window_size, window_size), num_heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        attn_mask = None
        if shift_size > 0:
            H, W = input_resolution
            img_mask = torch.zeros((1, H, W, 1))
            cnt = 0
            for h in (slice(0, -window_size), slice(-window_size, -shift_size), slice(-shift_size, None)):
                for w in (slice(0, -window_size), slice(-window_size, -shift_size), slice(-shift_size, None)):
                    img_mask[:, h, w, :] = cnt
                    cnt += 1
            mw = window_partition(img_mask, window_size).view(-1, window_size * window_size)
            attn_mask = mw.unsqueeze(1) - mw.unsqueeze(2)
            attn_mask = attn_mask.masked_fill(attn_mask != 0, float(-100.0)).masked_fill(attn_mask == 0, float(0.0))
        self.register_buffer("attn_mask", attn_mask)

    def forward(self, x):
        H, W = self.input_resolution
        B, L, C = x.shape
        shortcut = x
        x = self.norm1(x).view(B, H, W, C)
        if self.shift_size > 0:
            x = torch.roll(x, shifts=(-self.shift_size, -self.shift_size), dims=(1, 2))
        ws = self.window_size
        xw = window_partition(x, ws).view(-1, ws * ws, C)
        aw = self.attn(xw, mask=self.attn_mask).view(-1, ws, ws, C)
        x = window_reverse(aw, ws, H, W)
        if self.shift_size > 0:
            x = torch.roll(x, shifts=(self.shift_size, self.shift_size), dims=(1, 2))
        x = shortcut + x.view(B, H * W, C)
        return x + self.mlp(self.norm2(x))


class PatchMerging(nn.Module):
    """2x2 neighbourhood concat + LayerNorm + the `reduction` Linear (wrapped as qlinear_reduction,
    reference utils/net_wrap.py:42)."""

    def __init__(self, input_resolution, dim):
        super().__init__()
        self.input_resolution = input_resolution
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)

    def forward(self, x):
        H, W = self.input_resolution
        B, L, C = x.shape
        x = x.view(B, H, W, C)
        x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1).view(B, -1, 4 * C)
        return self.reduction(self.norm(x))


class SwinLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, downsample):
        super().__init__()
        self.blocks = nn.ModuleList([
            SwinBlock(dim, input_resolution, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2)
            for i in range(depth)])
        self.downsample = PatchMerging(input_resolution, dim) if downsample else None

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        return self.downsample(x) if self.downsample is not None else x


class SwinPatchEmbed(nn.Module):
    def __init__(self, img_size, patch, in_chans, dim):
        super().__init__()
        self.grid = (img_size // patch, img_size // patch)
        self.proj = nn.Conv2d(in_chans, dim, kernel_size=patch, stride=patch)
        self.norm = nn.LayerNorm(dim)

    def forward(self, x):
        return self.norm(self.proj(x).flatten(2).transpose(1, 2))


class SwinTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=4, in_chans=3, num_classes=1000, embed_dim=96, depths=(2, 2, 6, 2),
                 num_heads=(3, 6, 12, 24), window_size=7):
        super().__init__()
        self.patch_embed = SwinPatchEmbed(img_size, patch_size, in_chans, embed_dim)
        g = self.patch_embed.grid
        self.layers = nn.ModuleList([
            SwinLayer(embed_dim * 2 ** i, (g[0] // 2 ** i, g[1] // 2 ** i), depths[i], num_heads[i], window_size,
                      downsample=i < len(depths) - 1)
            for i in range(len(depths))])
        self.num_features = embed_dim * 2 ** (len(depths) - 1)
        self.norm = nn.LayerNorm(self.num_features)
        self.head = nn.Linear(self.num_features, num_classes)
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Conv2d)):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, WindowAttention):
                nn.init.trunc_normal_(m.relative_position_bias_table, std=0.02)

    def forward(self, x):
        x = self.patch_embed(x)
        for layer in self.layers:
            x = layer(x)
        x = self.norm(x).mean(dim=1)          # AdaptiveAvgPool1d(1) over the tokens
        return self.head(x)


_SWIN_ZOO = {
    # name: (img, window, dim, depths, heads)
    "swin_tiny_patch4_window7_224": (224, 7, 96, (2, 2, 6, 2), (3, 6, 12, 24)),
    "swin_small_patch4_window7_224": (224, 7, 96, (2, 2, 18, 2), (3, 6, 12, 24)),
    "swin_base_patch4_window7_224": (224, 7, 128, (2, 2, 18, 2), (4, 8, 16, 32)),
    "swin_large_patch4_window7_224": (224, 7, 192, (2, 2, 18, 2), (6, 12, 24, 48)),
    "swin_base_patch4_window12_384": (384, 12, 128, (2, 2, 18, 2), (4, 8, 16, 32)),
    "swin_large_patch4_window12_384": (384, 12, 192, (2, 2, 18, 2), (6, 12, 24, 48)),
}

_ZOO = {
    # name: (img, patch, dim, depth, heads)
    "vit_tiny_patch16_224": (224, 16, 192, 12, 3),
    "vit_small_patch32_224": (224, 32, 384, 12, 6),
    "vit_small_patch16_224": (224, 16, 384, 12, 6),
    "vit_base_patch16_224": (224, 16, 768, 12, 12),
    "vit_base_patch16_384": (384, 16, 768, 12, 12),
    "vit_base_patch32_224": (224, 32, 768, 12, 12),
    "vit_base_patch32_384": (384, 32, 768, 12, 12),
    "vit_large_patch16_224": (224, 16, 1024, 24, 16),
    "vit_large_patch16_384": (384, 16, 1024, 24, 16),
    "vit_large_patch32_224": (224, 32, 1024, 24, 16),
    "vit_large_patch32_384": (384, 32, 1024, 24, 16),
    "deit_tiny_patch16_224": (224, 16, 192, 12, 3),
    "deit_small_patch16_224": (224, 16, 384, 12, 6),
    "deit_base_patch16_224": (224, 16, 768, 12, 12),
    "deit_base_patch16_384": (384, 16, 768, 12, 12),
}


def input_size(name):
    """Image side the named model is defined for."""
    if name in _SWIN_ZOO:
        return _SWIN_ZOO[name][0]
    return _ZOO[name][0]


def get_net(name, seed=0, device=None, **overrides):
    """Build a ViT / DeiT / Swin by timm name (reference utils/models.py:62-91, without the pretrained download).

    Returns the net in eval mode, on the GPU when one is visible.
    """
    if name in _SWIN_ZOO:
        img, window, dim, depths, heads = _SWIN_ZOO[name]
        cfg = dict(img_size=img, window_size=window, embed_dim=dim, depths=depths, num_heads=heads)
        cls = SwinTransformer
    elif name in _ZOO:
        img, patch, dim, depth, heads = _ZOO[name]
        cfg = dict(img_size=img, patch_size=patch, embed_dim=dim, depth=depth, num_heads=heads)
        cls = VisionTransformer
    else:
        raise KeyError(f"unknown model {name}; known: {sorted(_ZOO) + sorted(_SWIN_ZOO)}")
    cfg.update(overrides)
    g = torch.random.get_rng_state()
    torch.manual_seed(seed)
    net = cls(**cfg)
    torch.random.set_rng_state(g)
    net.p4v_name = name                      # utils.datasets.data_config reads it (timm keeps this in net.default_cfg)
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    return net.to(device).eval()


def load_pretrained(net, path):
    """Load a timm checkpoint of the same architecture into a net built by `get_net` (the reference gets these weights from
    `timm.create_model(name, pretrained=True)`, utils/models.py:77; the parameter names here are timm's).

    `path`: a `.pth` / `.pt` / `.bin` state dict (optionally nested under "model" / "state_dict") or a `.safetensors` file.
    Every PARAMETER of the net must be present with its shape; recomputable buffers (relative_position_index, attn_mask) and
    checkpoint entries the net does not have are reported, not fatal -- except distillation heads (`head_dist`, `dist_token`):
    the reference's wrapper cannot map them either (utils/net_wrap.py:42,67 raises KeyError).  Returns (missing, unexpected)."""
    if str(path).endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(str(path))
    else:
        sd = torch.load(str(path), map_location="cpu", weights_only=True)
        for k in ("model", "state_dict"):
            if isinstance(sd, dict) and k in sd and isinstance(sd[k], dict):
                sd = sd[k]
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    if any(k.startswith(("head_dist", "dist_token")) for k in sd):
        raise KeyError("distilled DeiT checkpoint (head_dist / dist_token): not supported, as in the reference's wrap_modules_in_net")
    own = net.state_dict()
    params = {n for n, _ in net.named_parameters()}
    missing = [k for k in own if k not in sd]
    bad = [k for k in missing if k in params]
    if bad:
        raise KeyError(f"checkpoint {path} lacks parameters of {getattr(net, 'p4v_name', type(net).__name__)}: {bad[:8]}{' ...' if len(bad) > 8 else ''}")
    for k, v in sd.items():
        if k in own and tuple(v.shape) != tuple(own[k].shape):
            if v.numel() == own[k].numel():
                sd[k] = v.reshape(own[k].shape)        # e.g. a patch embedding stored as a flattened Linear
            else:
                raise ValueError(f"checkpoint {path}: {k} has shape {tuple(v.shape)}, the net expects {tuple(own[k].shape)}")
    unexpected = [k for k in sd if k not in own]
    net.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
    return missing, unexpected
