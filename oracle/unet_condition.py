"""Oracle restatement of diffusers==0.24.0 `UNet2DConditionModel` (TEST INFRASTRUCTURE ONLY).

The reference builds it at `scripts/train_unet.py:139-159` (block_out_channels (128,256,512,512), three
CrossAttnDownBlock2D + DownBlock2D, UpBlock2D + three CrossAttnUpBlock2D, `cross_attention_dim` = width of the audio
encoding) and calls it as `self.unet(images, t, encoding)["sample"]` (`audiodiffusion/pipeline_audio_diffusion.py:160-161`,
`scripts/train_unet.py:254-255`), `encoding` of shape (batch, seq_length, cross_attention_dim) (`:107`).  The arithmetic
lives in diffusers (models/unet_2d_condition.py, unet_2d_blocks.py, transformer_2d.py, attention.py,
attention_processor.py), not vendored; this file restates it in plain torch-CPU fp32 with the same state-dict key names.

Library defaults the reference relies on [3P-recall, diffusers 0.24.0]: `attention_head_dim=8` is READ AS THE NUMBER OF
HEADS (`num_attention_heads = num_attention_heads or attention_head_dim`), so a block of C channels runs 8 heads of C/8;
`transformer_layers_per_block=1`; `use_linear_projection=False` (proj_in / proj_out are 1x1 convolutions);
Transformer2DModel's GroupNorm(32, eps=1e-6); BasicTransformerBlock = LayerNorm -> self-attention (q/k/v without bias,
to_out.0 with bias) -> LayerNorm -> cross-attention on the encoding -> LayerNorm -> GEGLU feed-forward (inner 4C), each
with a residual; `mid_block_type="UNetMidBlock2DCrossAttn"`; everything else as UNet2DModel (oracle/unet.py).

Parity unpinned (see oracle/__init__.py): the reference holds no golden vectors for this model.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .unet import Downsample2D, ResnetBlock2D, TimestepEmbedding, Upsample2D, timestep_embedding

DEFAULT_CONFIG = dict(  # scripts/train_unet.py:139-159 + diffusers 0.24.0 defaults
    sample_size=64,
    in_channels=1,
    out_channels=1,
    layers_per_block=2,
    block_out_channels=(128, 256, 512, 512),
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    cross_attention_dim=100,
    attention_head_dim=8,       # = number of heads (see the module docstring)
    norm_num_groups=32,
    norm_eps=1e-5,
    freq_shift=0,
    flip_sin_to_cos=True,
)


class CrossAttention(nn.Module):
    """attention_processor.Attention: heads x dim_head, q/k/v without bias, `to_out.0` with bias."""

    def __init__(self, query_dim, heads, dim_head, cross_dim=None):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.scale = heads, dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(cross_dim or query_dim, inner, bias=False)
        self.to_v = nn.Linear(cross_dim or query_dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim)])

    def forward(self, x, context=None):
        context = x if context is None else context
        B, T, _ = x.shape

        def split(t):
            return t.view(B, t.shape[1], self.heads, -1).transpose(1, 2)       # (B, heads, T, d)

        q, k, v = split(self.to_q(x)), split(self.to_k(context)), split(self.to_v(context))
        p = torch.softmax(q @ k.transpose(-1, -2) * self.scale, dim=-1)
        o = (p @ v).transpose(1, 2).reshape(B, T, -1)
        return self.to_out[0](o)


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Dropout(0.0), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = CrossAttention(dim, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = CrossAttention(dim, heads, dim_head, cross_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, context):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), context) + x
        return self.ff(self.norm3(x)) + x


class Transformer2DModel(nn.Module):
    def __init__(self, ch, heads, cross_dim, groups):
        super().__init__()
        self.norm = nn.GroupNorm(groups, ch, eps=1e-6)
        self.proj_in = nn.Conv2d(ch, ch, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(ch, heads, ch // heads, cross_dim)])
        self.proj_out = nn.Conv2d(ch, ch, 1)

    def forward(self, x, context):
        B, C, H, W = x.shape
        h = self.proj_in(self.norm(x))
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
        for blk in self.transformer_blocks:
            h = blk(h, context)
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        return self.proj_out(h) + x


class DownBlock(nn.Module):
    def __init__(self, cin, cout, temb_ch, n_layers, cross, add_down, cfg):
        super().__init__()
        g, eps = cfg["norm_num_groups"], cfg["norm_eps"]
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb_ch, g, eps) for i in range(n_layers)])
        self.attentions = nn.ModuleList([Transformer2DModel(cout, cfg["attention_head_dim"], cfg["cross_attention_dim"], g)
                                         for _ in range(n_layers)]) if cross else None
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_down else None

    def forward(self, h, temb, context):
        outs = ()
        for i, r in enumerate(self.resnets):
            h = r(h, temb)
            if self.attentions is not None:
                h = self.attentions[i](h, context)
            outs += (h,)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
            outs += (h,)
        return h, outs


class UpBlock(nn.Module):
    def __init__(self, cin, prev, cout, temb_ch, n_layers, cross, add_up, cfg):
        super().__init__()
        g, eps = cfg["norm_num_groups"], cfg["norm_eps"]
        rs = []
        for i in range(n_layers):
            skip = cin if i == n_layers - 1 else cout
            rin = prev if i == 0 else cout
            rs.append(ResnetBlock2D(rin + skip, cout, temb_ch, g, eps))
        self.resnets = nn.ModuleList(rs)
        self.attentions = nn.ModuleList([Transformer2DModel(cout, cfg["attention_head_dim"], cfg["cross_attention_dim"], g)
                                         for _ in range(n_layers)]) if cross else None
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, h, skips, temb, context):
        for i, r in enumerate(self.resnets):
            s = skips[-1]
            skips = skips[:-1]
            h = r(torch.cat([h, s], dim=1), temb)
            if self.attentions is not None:
                h = self.attentions[i](h, context)
        if self.upsamplers is not None:
            h = self.upsamplers[0](h)
        return h


class MidBlock(nn.Module):
    def __init__(self, ch, temb_ch, cfg):
        super().__init__()
        g, eps = cfg["norm_num_groups"], cfg["norm_eps"]
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb_ch, g, eps) for _ in range(2)])
        self.attentions = nn.ModuleList([Transformer2DModel(ch, cfg["attention_head_dim"], cfg["cross_attention_dim"], g)])

    def forward(self, h, temb, context):
        h = self.resnets[0](h, temb)
        h = self.attentions[0](h, context)
        return self.resnets[1](h, temb)


class UNet2DConditionModel(nn.Module):
    """`forward(sample, timestep, encoder_hidden_states)` returns {"sample": ...} (`pipeline_audio_diffusion.py:161`)."""

    def __init__(self, **kw):
        super().__init__()
        cfg = dict(DEFAULT_CONFIG)
        cfg.update(kw)
        self.config = cfg
        boc = tuple(cfg["block_out_channels"])
        temb_ch = boc[0] * 4
        self.sample_size = cfg["sample_size"]
        self.in_channels = cfg["in_channels"]
        self.conv_in = nn.Conv2d(cfg["in_channels"], boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb_ch)
        L = cfg["layers_per_block"]
        self.down_blocks = nn.ModuleList()
        out = boc[0]
        for i, t in enumerate(cfg["down_block_types"]):
            cin, out = out, boc[i]
            self.down_blocks.append(DownBlock(cin, out, temb_ch, L, t.startswith("CrossAttn"), i != len(boc) - 1, cfg))
        self.mid_block = MidBlock(boc[-1], temb_ch, cfg)
        self.up_blocks = nn.ModuleList()
        rev = boc[::-1]
        out = rev[0]
        for i, t in enumerate(cfg["up_block_types"]):
            prev, out = out, rev[i]
            cin = rev[min(i + 1, len(boc) - 1)]
            self.up_blocks.append(UpBlock(cin, prev, out, temb_ch, L + 1, t.startswith("CrossAttn"), i != len(boc) - 1, cfg))
        self.conv_norm_out = nn.GroupNorm(cfg["norm_num_groups"], boc[0], eps=cfg["norm_eps"])
        self.conv_out = nn.Conv2d(boc[0], cfg["out_channels"], 3, padding=1)

    def forward(self, sample, timestep, encoder_hidden_states):
        cfg = self.config
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.long)
        elif t.dim() == 0:
            t = t[None]
        t = t * torch.ones(sample.shape[0], dtype=t.dtype)
        temb = timestep_embedding(t, cfg["block_out_channels"][0], cfg["flip_sin_to_cos"], cfg["freq_shift"])
        emb = self.time_embedding(temb.to(sample.dtype))
        ctx = encoder_hidden_states
        h = self.conv_in(sample)
        skips = (h,)
        for blk in self.down_blocks:
            h, outs = blk(h, emb, ctx)
            skips += outs
        h = self.mid_block(h, emb, ctx)
        for blk in self.up_blocks:
            n = len(blk.resnets)
            s, skips = skips[-n:], skips[:-n]
            h = blk(h, s, emb, ctx)
        h = self.conv_out(F.silu(self.conv_norm_out(h)))
        return {"sample": h}
