"""Oracle restatement of diffusers==0.24.0 `AutoencoderKL` (TEST INFRASTRUCTURE ONLY).

Used by the reference for latent audio diffusion: `vqvae.encode(x).latent_dist.sample(generator)`
(`audiodiffusion/pipeline_audio_diffusion.py:144`, `scripts/train_unet.py:104,233`) and
`vqvae.decode(z)["sample"]` (`pipeline_audio_diffusion.py:190`). Shape spec: `config/ldm_autoencoder_kl.yaml:18-28`
through `audiodiffusion/utils.py:132-153` (block_out_channels (128,256,512,512), layers_per_block 2,
latent_channels 1, in/out 1, Down/UpDecoderBlock2D x4); state-dict keys as `utils.py:156-291` emits them.
diffusers is not vendored; this restates models/autoencoder_kl.py + vae.py + unet_2d_blocks.py (SURVEY.md §8(a)
rows V1-V3): GroupNorm eps 1e-6, asymmetric (0,1,0,1) zero pad before the stride-2 convs, single-head attention
(head_dim = channels) in the mid blocks, DiagonalGaussian posterior with logvar clamped to [-30, 20].
Parity unpinned (see oracle/__init__.py).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .unet import Attention, Downsample2D, ResnetBlock2D, Upsample2D

DEFAULT_CONFIG = dict(  # audiodiffusion/utils.py:132-153 applied to config/ldm_autoencoder_kl.yaml:18-28
    sample_size=(256, 256), in_channels=1, out_channels=1, latent_channels=1, layers_per_block=2,
    block_out_channels=(128, 256, 512, 512),
    down_block_types=("DownEncoderBlock2D",) * 4, up_block_types=("UpDecoderBlock2D",) * 4,
    norm_num_groups=32, scaling_factor=0.18215,
)
EPS = 1e-6


class _Mid(nn.Module):
    def __init__(self, ch, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, None, groups, EPS) for _ in range(2)])
        self.attentions = nn.ModuleList([Attention(ch, ch, groups, EPS)])  # heads = ch // ch = 1

    def forward(self, h):
        h = self.resnets[0](h, None)
        h = self.attentions[0](h)
        return self.resnets[1](h, None)


class _EncBlock(nn.Module):
    def __init__(self, cin, cout, n, groups, down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, None, groups, EPS) for i in range(n)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout, padding=0)]) if down else None

    def forward(self, h):
        for r in self.resnets:
            h = r(h, None)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
        return h


class _DecBlock(nn.Module):
    def __init__(self, cin, cout, n, groups, up):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, None, groups, EPS) for i in range(n)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if up else None

    def forward(self, h):
        for r in self.resnets:
            h = r(h, None)
        if self.upsamplers is not None:
            h = self.upsamplers[0](h)
        return h


class Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        boc, g = cfg["block_out_channels"], cfg["norm_num_groups"]
        self.conv_in = nn.Conv2d(cfg["in_channels"], boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        out = boc[0]
        for i in range(len(boc)):
            cin, out = out, boc[i]
            self.down_blocks.append(_EncBlock(cin, out, cfg["layers_per_block"], g, i != len(boc) - 1))
        self.mid_block = _Mid(boc[-1], g)
        self.conv_norm_out = nn.GroupNorm(g, boc[-1], eps=EPS)
        self.conv_out = nn.Conv2d(boc[-1], 2 * cfg["latent_channels"], 3, padding=1)

    def forward(self, x):
        h = self.conv_in(x)
        for b in self.down_blocks:
            h = b(h)
        h = self.mid_block(h)
        return self.conv_out(F.silu(self.conv_norm_out(h)))


class Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        boc, g = cfg["block_out_channels"], cfg["norm_num_groups"]
        self.conv_in = nn.Conv2d(cfg["latent_channels"], boc[-1], 3, padding=1)
        self.mid_block = _Mid(boc[-1], g)
        self.up_blocks = nn.ModuleList()
        rev = boc[::-1]
        out = rev[0]
        for i in range(len(boc)):
            prev, out = out, rev[i]
            self.up_blocks.append(_DecBlock(prev, out, cfg["layers_per_block"] + 1, g, i != len(boc) - 1))
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=EPS)
        self.conv_out = nn.Conv2d(boc[0], cfg["out_channels"], 3, padding=1)

    def forward(self, z):
        h = self.conv_in(z)
        h = self.mid_block(h)
        for b in self.up_blocks:
            h = b(h)
        return self.conv_out(F.silu(self.conv_norm_out(h)))


class DiagonalGaussianDistribution:
    def __init__(self, parameters):
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None, noise=None):
        if noise is None:
            noise = torch.randn(self.mean.shape, generator=generator, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class _EncodeOutput:
    def __init__(self, dist):
        self.latent_dist = dist


class AutoencoderKL(nn.Module):
    def __init__(self, **kw):
        super().__init__()
        cfg = dict(DEFAULT_CONFIG)
        cfg.update(kw)
        cfg["block_out_channels"] = tuple(cfg["block_out_channels"])
        self.config = cfg
        self.encoder = Encoder(cfg)
        self.decoder = Decoder(cfg)
        self.quant_conv = nn.Conv2d(2 * cfg["latent_channels"], 2 * cfg["latent_channels"], 1)
        self.post_quant_conv = nn.Conv2d(cfg["latent_channels"], cfg["latent_channels"], 1)

    def encode(self, x):
        return _EncodeOutput(DiagonalGaussianDistribution(self.quant_conv(self.encoder(x))))

    def decode(self, z):
        return {"sample": self.decoder(self.post_quant_conv(z))}
