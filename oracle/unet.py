"""Oracle restatement of diffusers==0.24.0 `UNet2DModel` (TEST INFRASTRUCTURE ONLY).

The reference constructs it at `scripts/train_unet.py:115-137` and calls it at
`audiodiffusion/pipeline_audio_diffusion.py:161-163,237`. The arithmetic lives in
diffusers (models/unet_2d.py, unet_2d_blocks.py, resnet.py, attention_processor.py,
embeddings.py) which is not vendored; this file restates it op-for-op in plain
torch-CPU fp32 with the same module / state-dict key names
(SURVEY.md §8(a) rows U1-U8, §8(b) "On-disk format").

Parity unpinned (see oracle/__init__.py); analytic anchor: the
train_unet.py:115-137 config with 1 in/out channel has 113 668 609 parameters.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

DEFAULT_CONFIG = dict(  # scripts/train_unet.py:115-137 + diffusers 0.24.0 defaults
    sample_size=256,
    in_channels=1,
    out_channels=1,
    layers_per_block=2,
    block_out_channels=(128, 128, 256, 256, 512, 512),
    down_block_types=("DownBlock2D", "DownBlock2D", "DownBlock2D", "DownBlock2D", "AttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "AttnUpBlock2D", "UpBlock2D", "UpBlock2D", "UpBlock2D", "UpBlock2D"),
    attention_head_dim=8,
    norm_num_groups=32,
    norm_eps=1e-5,
    freq_shift=0,
    flip_sin_to_cos=True,
)


def timestep_embedding(timesteps, dim, flip_sin_to_cos=True, freq_shift=0, max_period=10000):
    """diffusers embeddings.get_timestep_embedding (row U1)."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32)
    exponent = exponent / (half - freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, cemb):
        super().__init__()
        self.linear_1 = nn.Linear(cin, cemb)
        self.linear_2 = nn.Linear(cemb, cemb)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    """diffusers resnet.ResnetBlock2D, time_embedding_norm="default", output_scale_factor=1 (row U3)."""

    def __init__(self, cin, cout, temb_ch, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_ch, cout) if temb_ch is not None else None
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        if self.time_emb_proj is not None:
            h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return (x + h) / 1.0


class Attention(nn.Module):
    """diffusers Attention built `_from_deprecated_attn_block` (row U6): GN -> q,k,v -> heads -> softmax -> out + residual."""

    def __init__(self, ch, head_dim, groups, eps):
        super().__init__()
        self.heads = ch // head_dim if head_dim is not None else 1
        self.group_norm = nn.GroupNorm(groups, ch, eps=eps)
        self.to_q = nn.Linear(ch, ch)
        self.to_k = nn.Linear(ch, ch)
        self.to_v = nn.Linear(ch, ch)
        self.to_out = nn.ModuleList([nn.Linear(ch, ch), nn.Dropout(0.0)])

    def forward(self, x):
        b, c, h, w = x.shape
        res = x
        hs = x.view(b, c, h * w).transpose(1, 2)
        hs = self.group_norm(hs.transpose(1, 2)).transpose(1, 2)
        q, k, v = self.to_q(hs), self.to_k(hs), self.to_v(hs)
        d = c // self.heads

        def split(t):
            return t.view(b, -1, self.heads, d).transpose(1, 2)

        q, k, v = split(q), split(k), split(v)
        scores = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
        probs = scores.float().softmax(dim=-1)
        o = torch.matmul(probs, v).transpose(1, 2).reshape(b, -1, c)
        o = self.to_out[0](o)
        o = o.transpose(-1, -2).reshape(b, c, h, w)
        return (o + res) / 1.0


class Downsample2D(nn.Module):
    def __init__(self, ch, padding=1):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=padding)

    def forward(self, x):
        if self.padding == 0:  # AutoencoderKL encoder: asymmetric (0,1,0,1) zero pad
            x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    def __init__(self, cin, cout, temb_ch, n_layers, attn, add_down, cfg):
        super().__init__()
        g, eps = cfg["norm_num_groups"], cfg["norm_eps"]
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, temb_ch, g, eps) for i in range(n_layers)]
        )
        if attn:
            self.attentions = nn.ModuleList([Attention(cout, cfg["attention_head_dim"], g, eps) for _ in range(n_layers)])
        else:
            self.attentions = None
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_down else None

    def forward(self, h, temb):
        outs = ()
        for i, r in enumerate(self.resnets):
            h = r(h, temb)
            if self.attentions is not None:
                h = self.attentions[i](h)
            outs += (h,)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
            outs += (h,)
        return h, outs


class UpBlock(nn.Module):
    def __init__(self, cin, prev, cout, temb_ch, n_layers, attn, add_up, cfg):
        super().__init__()
        g, eps = cfg["norm_num_groups"], cfg["norm_eps"]
        rs = []
        for i in range(n_layers):
            skip = cin if i == n_layers - 1 else cout
            rin = prev if i == 0 else cout
            rs.append(ResnetBlock2D(rin + skip, cout, temb_ch, g, eps))
        self.resnets = nn.ModuleList(rs)
        if attn:
            self.attentions = nn.ModuleList([Attention(cout, cfg["attention_head_dim"], g, eps) for _ in range(n_layers)])
        else:
            self.attentions = None
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, h, skips, temb):
        for i, r in enumerate(self.resnets):
            s = skips[-1]
            skips = skips[:-1]
            h = r(torch.cat([h, s], dim=1), temb)
            if self.attentions is not None:
                h = self.attentions[i](h)
        if self.upsamplers is not None:
            h = self.upsamplers[0](h)
        return h


class MidBlock(nn.Module):
    def __init__(self, ch, temb_ch, cfg, head_dim):
        super().__init__()
        g, eps = cfg["norm_num_groups"], cfg["norm_eps"]
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb_ch, g, eps) for _ in range(2)])
        self.attentions = nn.ModuleList([Attention(ch, head_dim, g, eps)])

    def forward(self, h, temb):
        h = self.resnets[0](h, temb)
        h = self.attentions[0](h)
        return self.resnets[1](h, temb)


class UNet2DModel(nn.Module):
    """Oracle UNet2DModel. `forward(sample, timestep)` returns {"sample": ...} as the
    reference accesses it (`pipeline_audio_diffusion.py:163`)."""

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
            self.down_blocks.append(DownBlock(cin, out, temb_ch, L, t.startswith("Attn"), i != len(boc) - 1, cfg))
        self.mid_block = MidBlock(boc[-1], temb_ch, cfg, cfg["attention_head_dim"])
        self.up_blocks = nn.ModuleList()
        rev = boc[::-1]
        out = rev[0]
        for i, t in enumerate(cfg["up_block_types"]):
            prev, out = out, rev[i]
            cin = rev[min(i + 1, len(boc) - 1)]
            self.up_blocks.append(UpBlock(cin, prev, out, temb_ch, L + 1, t.startswith("Attn"), i != len(boc) - 1, cfg))
        self.conv_norm_out = nn.GroupNorm(cfg["norm_num_groups"], boc[0], eps=cfg["norm_eps"])
        self.conv_out = nn.Conv2d(boc[0], cfg["out_channels"], 3, padding=1)

    def forward(self, sample, timestep):
        cfg = self.config
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.long)
        elif t.dim() == 0:
            t = t[None]
        t = t * torch.ones(sample.shape[0], dtype=t.dtype)
        temb = timestep_embedding(t, cfg["block_out_channels"][0], cfg["flip_sin_to_cos"], cfg["freq_shift"])
        emb = self.time_embedding(temb.to(sample.dtype))
        h = self.conv_in(sample)
        skips = (h,)
        for blk in self.down_blocks:
            h, outs = blk(h, emb)
            skips += outs
        h = self.mid_block(h, emb)
        for blk in self.up_blocks:
            n = len(blk.resnets)
            s, skips = skips[-n:], skips[:-n]
            h = blk(h, s, emb)
        h = self.conv_out(F.silu(self.conv_norm_out(h)))
        return {"sample": h}


def remap_deprecated_attention_keys(sd):
    """2022-era checkpoints name attention params query/key/value/proj_attn
    (`audiodiffusion/utils.py:41-54`); diffusers 0.24 maps them on load."""
    m = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}
    out = {}
    for k, v in sd.items():
        parts = k.split(".")
        if "attentions" in parts and parts[-2] in m:
            parts[-2] = m[parts[-2]]
            k = ".".join(parts)
        out[k] = v
    return out
