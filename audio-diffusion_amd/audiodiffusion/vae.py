"""AutoencoderKL drop-in backed by the native executor (csrc/vae_exec.hip) — latent audio diffusion (config 4).

Duck-types what the reference touches on `pipeline.vqvae`:
`vqvae.encode(x).latent_dist.sample(generator=)` (`pipeline_audio_diffusion.py:144`, `train_unet.py:104,233`),
`vqvae.decode(z)["sample"]` (`pipeline_audio_diffusion.py:190`), `vqvae.config["latent_channels"]`
(`train_unet.py:81,117`), and the diffusers on-disk layout `vqvae/config.json` + weights with the key names
`audiodiffusion/utils.py:156-291` emits (deprecated attention names accepted).
"""
import ctypes as C
import json
import math
import os

import torch

from . import _native as N
from .schedulers import FrozenConfig, randn_tensor

_DEFAULTS = dict(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",),
                 up_block_types=("UpDecoderBlock2D",), block_out_channels=(64,), layers_per_block=1, act_fn="silu",
                 latent_channels=4, norm_num_groups=32, sample_size=32, scaling_factor=0.18215, force_upcast=True)


def param_specs(cfg):
    boc = list(cfg["block_out_channels"])
    nb, L, cz = len(boc), cfg["layers_per_block"], cfg["latent_channels"]
    out = []

    def conv(p, co, ci, ks):
        out.append((p + ".weight", (co, ci, ks, ks), ci * ks * ks)), out.append((p + ".bias", (co,), ci * ks * ks))

    def lin(p, co, ci):
        out.append((p + ".weight", (co, ci), ci)), out.append((p + ".bias", (co,), ci))

    def gn(p, c):
        out.append((p + ".weight", (c,), 0)), out.append((p + ".bias", (c,), -1))

    def resnet(p, ci, co):
        gn(p + ".norm1", ci), conv(p + ".conv1", co, ci, 3), gn(p + ".norm2", co), conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".conv_shortcut", co, ci, 1)

    def mid(p, c):
        resnet(p + ".resnets.0", c, c)
        gn(p + ".attentions.0.group_norm", c)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            lin(p + ".attentions.0." + n, c, c)
        resnet(p + ".resnets.1", c, c)

    conv("encoder.conv_in", boc[0], cfg["in_channels"], 3)
    o = boc[0]
    for i in range(nb):
        ci, o = o, boc[i]
        for j in range(L):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", ci if j == 0 else o, o)
        if i != nb - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", o, o, 3)
    mid("encoder.mid_block", boc[-1])
    gn("encoder.conv_norm_out", boc[-1]), conv("encoder.conv_out", 2 * cz, boc[-1], 3)
    conv("quant_conv", 2 * cz, 2 * cz, 1), conv("post_quant_conv", cz, cz, 1)
    conv("decoder.conv_in", boc[-1], cz, 3)
    mid("decoder.mid_block", boc[-1])
    rev = boc[::-1]
    o = rev[0]
    for i in range(nb):
        prev, o = o, rev[i]
        for j in range(L + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else o, o)
        if i != nb - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", o, o, 3)
    gn("decoder.conv_norm_out", boc[0]), conv("decoder.conv_out", cfg["out_channels"], boc[0], 3)
    return out


_OLD_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def _canon_key(k):
    parts = k.split(".")
    if "attentions" in parts and parts[-2] in _OLD_ATTN:
        parts[-2] = _OLD_ATTN[parts[-2]]
    return ".".join(parts)


class VaeConfigStruct(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("out_channels", C.c_int), ("latent_channels", C.c_int),
                ("layers_per_block", C.c_int), ("n_blocks", C.c_int), ("block_out_channels", C.c_int * 8),
                ("norm_num_groups", C.c_int), ("sample_h", C.c_int), ("sample_w", C.c_int)]


class DiagonalGaussianDistribution:
    """`latent_dist` of `vqvae.encode(...)`: the moments stay on the device; `.sample()` runs the fused kernel."""

    def __init__(self, vae, x):
        self._vae, self._x = vae, x

    def sample(self, generator=None, noise=None):
        v = self._vae
        B = self._x.shape[0]
        lh, lw = v.latent_size(tuple(self._x.shape[2:]))
        shape = (B, v.config.latent_channels, lh, lw)
        if noise is None:
            noise = randn_tensor(shape, generator, self._x.device, torch.float32)
        return v._encode(self._x, noise.contiguous(), 1.0)

    def mode(self):
        return self._vae._encode(self._x, None, 1.0)


class EncoderOutput:
    def __init__(self, dist):
        self.latent_dist = dist


class DecoderOutput(dict):
    __getattr__ = dict.__getitem__


class AutoencoderKL:
    config_name = "config.json"

    def __init__(self, **kwargs):
        cfg = dict(_DEFAULTS)
        cfg.update({k: v for k, v in kwargs.items() if not k.startswith("_")})
        for k in ("down_block_types", "up_block_types", "block_out_channels"):
            cfg[k] = tuple(cfg[k])
        if any(t != "DownEncoderBlock2D" for t in cfg["down_block_types"]) or \
                any(t != "UpDecoderBlock2D" for t in cfg["up_block_types"]):
            raise NotImplementedError("only DownEncoderBlock2D / UpDecoderBlock2D are implemented")
        if cfg["act_fn"] != "silu":
            raise NotImplementedError("only act_fn='silu' is implemented")
        self.config = FrozenConfig(cfg)
        self.dtype = torch.float32
        self._handle, self._handle_hw = None, None
        self._sd = {}

    def _hw(self, hw=None):
        if hw is not None:
            return tuple(hw)
        ss = self.config.sample_size
        return (ss, ss) if isinstance(ss, int) else tuple(ss)

    def _ensure_handle(self, hw):
        hw = tuple(hw)
        if self._handle is not None and self._handle_hw == hw:
            return self._handle
        self._free()
        c = self.config
        nc = VaeConfigStruct(c.in_channels, c.out_channels, c.latent_channels, c.layers_per_block,
                             len(c.block_out_channels))
        for i, v in enumerate(c.block_out_channels):
            nc.block_out_channels[i] = v
        nc.norm_num_groups = c.norm_num_groups
        nc.sample_h, nc.sample_w = hw
        h = C.c_void_p()
        lib = N.lib()
        N.check(lib.adm_vae_create(C.cast(C.byref(nc), C.c_void_p), C.byref(h)))
        self._handle, self._handle_hw = h, hw
        for k, t in self._sd.items():
            N.check(lib.adm_vae_set_param(h, k.encode(), C.c_void_p(t.data_ptr()), t.numel()))
        return h

    def _free(self):
        if getattr(self, "_handle", None) is not None:
            N.lib().adm_vae_destroy(self._handle)
        self._handle = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass

    def latent_size(self, hw=None):
        h = self._ensure_handle(self._hw(hw))
        a, b = C.c_int(), C.c_int()
        N.check(N.lib().adm_vae_latent_dims(h, C.byref(a), C.byref(b)))
        return a.value, b.value

    # ---- weights -------------------------------------------------------------------------------------------
    def load_state_dict(self, sd, strict=True):
        specs = {k: s for k, s, _ in param_specs(self.config)}
        new = {}
        for k, v in sd.items():
            ck = _canon_key(k)
            if ck not in specs:
                if strict:
                    raise KeyError(f"unexpected key {k}")
                continue
            if tuple(v.shape) != tuple(specs[ck]) and v.numel() == math.prod(specs[ck]):
                v = v.reshape(specs[ck])  # conv_attn_to_linear (audiodiffusion/utils.py:117-129)
            if tuple(v.shape) != tuple(specs[ck]):
                raise ValueError(f"shape mismatch for {k}")
            new[ck] = v.detach().to(torch.float32).cpu().contiguous()
        missing = [k for k in specs if k not in new]
        if strict and missing:
            raise KeyError(f"missing keys: {missing[:8]}")
        self._sd.update(new)
        self._free()
        return self

    def state_dict(self):
        return dict(self._sd)

    def init_random(self, seed=0):
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for k, shape, fan_in in param_specs(self.config):
            if fan_in == 0:
                sd[k] = torch.ones(shape)
            elif fan_in == -1:
                sd[k] = torch.zeros(shape)
            else:
                sd[k] = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
        return self.load_state_dict(sd)

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    # ---- reference members -----------------------------------------------------------------------------------
    def _encode(self, x, noise, scale):
        x = x.contiguous()
        h = self._ensure_handle(x.shape[2:])
        B = x.shape[0]
        lh, lw = self.latent_size(x.shape[2:])
        z = torch.empty((B, self.config.latent_channels, lh, lw), dtype=torch.float32, device=x.device)
        N.check(N.lib().adm_vae_encode(h, N.ptr(x), N.ptr(noise), C.c_float(scale), N.ptr(z), None, B, N.stream_for(x)))
        return z

    def encode(self, x, return_dict=True):
        assert x.dim() == 4 and x.dtype == torch.float32
        return EncoderOutput(DiagonalGaussianDistribution(self, x))

    def decode(self, z, return_dict=True, _in_scale=1.0):
        z = z.contiguous()
        B = z.shape[0]
        f = 2 ** (len(self.config.block_out_channels) - 1)
        hw = (z.shape[2] * f, z.shape[3] * f)
        h = self._ensure_handle(hw)
        out = torch.empty((B, self.config.out_channels) + hw, dtype=torch.float32, device=z.device)
        N.check(N.lib().adm_vae_decode(h, N.ptr(z), C.c_float(_in_scale), N.ptr(out), B, N.stream_for(z)))
        return DecoderOutput(sample=out)

    # ---- diffusers on-disk layout ------------------------------------------------------------------------------
    @classmethod
    def from_config(cls, cfg):
        return cls(**{k: v for k, v in dict(cfg).items() if not k.startswith("_")})

    @classmethod
    def from_pretrained(cls, path, subfolder=None):
        p = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(p, cls.config_name)) as f:
            m = cls.from_config(json.load(f))
        st = os.path.join(p, "diffusion_pytorch_model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(p, "diffusion_pytorch_model.bin"), map_location="cpu", weights_only=True)
        return m.load_state_dict(sd)

    def save_pretrained(self, path, safe_serialization=True):
        os.makedirs(path, exist_ok=True)
        d = {"_class_name": "AutoencoderKL", "_diffusers_version": "0.24.0"}
        d.update({k: (list(v) if isinstance(v, tuple) else v) for k, v in self.config.items()})
        with open(os.path.join(path, self.config_name), "w") as f:
            json.dump(d, f, indent=2, sort_keys=True)
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(self._sd, os.path.join(path, "diffusion_pytorch_model.safetensors"))
        else:
            torch.save(self._sd, os.path.join(path, "diffusion_pytorch_model.bin"))
