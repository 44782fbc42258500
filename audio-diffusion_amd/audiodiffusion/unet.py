"""UNet2DModel drop-in backed by the native executor (csrc/unet_exec.hip).

Duck-types what the reference touches on `pipeline.unet`:
`unet(images, t)["sample"]` (`pipeline_audio_diffusion.py:163,237`, `train_unet.py:257`), `.sample_size`
(assigned to at `:119`), `.in_channels` (`:124`), `.config`, and the diffusers on-disk layout
(`unet/config.json` + `diffusion_pytorch_model.{safetensors,bin}`, state-dict keys of diffusers==0.24.0,
including the 2022-era attention names `query/key/value/proj_attn`, `audiodiffusion/utils.py:41-54`).
The constructor arguments / defaults are those of `scripts/train_unet.py:115-137`.
"""
import ctypes as C
import json
import math
import os

import torch

from . import _native as N
from .schedulers import FrozenConfig

_DEFAULTS = dict(
    sample_size=None, in_channels=3, out_channels=3, center_input_sample=False, time_embedding_type="positional",
    freq_shift=0, flip_sin_to_cos=True,
    down_block_types=("DownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D"),
    up_block_types=("AttnUpBlock2D", "AttnUpBlock2D", "AttnUpBlock2D", "UpBlock2D"),
    block_out_channels=(224, 448, 672, 896), layers_per_block=2, mid_block_scale_factor=1, downsample_padding=1,
    downsample_type="conv", upsample_type="conv", dropout=0.0, act_fn="silu", attention_head_dim=8,
    norm_num_groups=32, attn_norm_num_groups=None, norm_eps=1e-5, resnet_time_scale_shift="default",
    add_attention=True, class_embed_type=None, num_class_embeds=None, num_train_timesteps=None,
)


class UNetOutput(dict):
    __getattr__ = dict.__getitem__


def param_specs(cfg):
    """(key, shape, fan_in) for every parameter, in diffusers' module order (mirrors csrc/unet_exec.hip declare_all)."""
    boc = list(cfg["block_out_channels"])
    nb, L = len(boc), cfg["layers_per_block"]
    temb = boc[0] * 4
    out = []
    cross = cfg.get("cross_attention_dim") or 0

    def conv(p, co, ci, ks):
        out.append((p + ".weight", (co, ci, ks, ks), ci * ks * ks))
        out.append((p + ".bias", (co,), ci * ks * ks))

    def lin(p, co, ci):
        out.append((p + ".weight", (co, ci), ci))
        out.append((p + ".bias", (co,), ci))

    def gn(p, c):
        out.append((p + ".weight", (c,), 0))
        out.append((p + ".bias", (c,), -1))

    def resnet(p, ci, co):
        gn(p + ".norm1", ci), conv(p + ".conv1", co, ci, 3), lin(p + ".time_emb_proj", co, temb)
        gn(p + ".norm2", co), conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".conv_shortcut", co, ci, 1)

    def attn(p, c):
        if cross:            # diffusers Transformer2DModel with one BasicTransformerBlock (UNet2DConditionModel)
            tb = p + ".transformer_blocks.0"
            gn(p + ".norm", c), conv(p + ".proj_in", c, c, 1)
            gn(tb + ".norm1", c)
            for n in ("attn1.to_q", "attn1.to_k", "attn1.to_v"):
                out.append((f"{tb}.{n}.weight", (c, c), c))
            lin(tb + ".attn1.to_out.0", c, c)
            gn(tb + ".norm2", c)
            out.append((tb + ".attn2.to_q.weight", (c, c), c))
            out.append((tb + ".attn2.to_k.weight", (c, cross), cross))
            out.append((tb + ".attn2.to_v.weight", (c, cross), cross))
            lin(tb + ".attn2.to_out.0", c, c)
            gn(tb + ".norm3", c)
            lin(tb + ".ff.net.0.proj", 8 * c, c), lin(tb + ".ff.net.2", c, 4 * c)
            conv(p + ".proj_out", c, c, 1)
            return
        gn(p + ".group_norm", c)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            lin(p + "." + n, c, c)

    conv("conv_in", boc[0], cfg["in_channels"], 3)
    lin("time_embedding.linear_1", temb, boc[0]), lin("time_embedding.linear_2", temb, temb)
    o = boc[0]
    for i, t in enumerate(cfg["down_block_types"]):
        ci, o = o, boc[i]
        for j in range(L):
            resnet(f"down_blocks.{i}.resnets.{j}", ci if j == 0 else o, o)
        if "Attn" in t:
            for j in range(L):
                attn(f"down_blocks.{i}.attentions.{j}", o)
        if i != nb - 1:
            conv(f"down_blocks.{i}.downsamplers.0.conv", o, o, 3)
    resnet("mid_block.resnets.0", boc[-1], boc[-1]), attn("mid_block.attentions.0", boc[-1])
    resnet("mid_block.resnets.1", boc[-1], boc[-1])
    rev = boc[::-1]
    o = rev[0]
    for i, t in enumerate(cfg["up_block_types"]):
        prev, o = o, rev[i]
        ci = rev[min(i + 1, nb - 1)]
        for j in range(L + 1):
            resnet(f"up_blocks.{i}.resnets.{j}", (prev if j == 0 else o) + (ci if j == L else o), o)
        if "Attn" in t:
            for j in range(L + 1):
                attn(f"up_blocks.{i}.attentions.{j}", o)
        if i != nb - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv", o, o, 3)
    gn("conv_norm_out", boc[0]), conv("conv_out", cfg["out_channels"], boc[0], 3)
    return out


_OLD_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def _canon_key(k):
    parts = k.split(".")
    if "attentions" in parts and parts[-2] in _OLD_ATTN:
        parts[-2] = _OLD_ATTN[parts[-2]]
    return ".".join(parts)


class UNet2DModel:
    config_name = "config.json"

    _defaults = _DEFAULTS

    def __init__(self, **kwargs):
        cfg = dict(self._defaults)
        cfg.update({k: v for k, v in kwargs.items() if not k.startswith("_")})
        for k in ("down_block_types", "up_block_types", "block_out_channels"):
            cfg[k] = tuple(cfg[k])
        self._validate(cfg)
        self.config = FrozenConfig(cfg)
        self.sample_size = cfg["sample_size"]
        self.in_channels = cfg["in_channels"]
        self.dtype = torch.float32
        self._handle = None
        self._handle_hw = None
        self._sd = {}
        self.device = N.default_device()

    @staticmethod
    def _validate(cfg):
        ok_down, ok_up = {"DownBlock2D", "AttnDownBlock2D"}, {"UpBlock2D", "AttnUpBlock2D"}
        bad = [t for t in cfg["down_block_types"] if t not in ok_down] + [t for t in cfg["up_block_types"] if t not in ok_up]
        if bad:
            raise NotImplementedError(f"block types {bad} are not implemented (hot path covers Down/AttnDown/Up/AttnUp)")
        for k, want in (("time_embedding_type", "positional"), ("act_fn", "silu"), ("resnet_time_scale_shift", "default"),
                        ("downsample_type", "conv"), ("upsample_type", "conv"), ("add_attention", True),
                        ("center_input_sample", False), ("class_embed_type", None), ("downsample_padding", 1),
                        ("mid_block_scale_factor", 1), ("attn_norm_num_groups", None)):
            if cfg.get(k, want) != want:
                raise NotImplementedError(f"UNet2DModel config {k}={cfg[k]!r} is not implemented (expected {want!r})")
        if len(cfg["block_out_channels"]) > 8:
            raise NotImplementedError("more than 8 blocks")

    # ---- native handle ---------------------------------------------------------------------------------
    def _hw(self):
        ss = self.sample_size
        return (ss, ss) if isinstance(ss, int) else tuple(ss)

    def _ensure_handle(self):
        hw = self._hw()
        if self._handle is not None and self._handle_hw == hw:
            return self._handle
        self._free()
        h = self._create_handle()
        for k, v in self._sd.items():
            self._upload(k, v)
        return h

    def _create_handle(self):
        hw = self._hw()
        c = self.config
        nc = N.UNetConfig()
        nc.in_channels, nc.out_channels = c.in_channels, c.out_channels
        nc.layers_per_block, nc.n_blocks = c.layers_per_block, len(c.block_out_channels)
        for i, v in enumerate(c.block_out_channels):
            nc.block_out_channels[i] = v
            nc.down_attn[i] = 2 if c.down_block_types[i].startswith("CrossAttn") else int(c.down_block_types[i].startswith("Attn"))
            nc.up_attn[i] = 2 if c.up_block_types[i].startswith("CrossAttn") else int(c.up_block_types[i].startswith("Attn"))
        nc.cross_attention_dim = int(c.get("cross_attention_dim") or 0)
        nc.attention_head_dim = c.attention_head_dim if c.attention_head_dim is not None else 0
        nc.norm_num_groups, nc.norm_eps = c.norm_num_groups, c.norm_eps
        nc.flip_sin_to_cos, nc.freq_shift = int(c.flip_sin_to_cos), float(c.freq_shift)
        nc.sample_h, nc.sample_w = hw
        h = C.c_void_p()
        N.check(N.lib().adm_unet_create(C.byref(nc), C.byref(h)))
        self._handle, self._handle_hw = h, hw
        self._loss_scale = 1.0          # a fresh native handle starts at loss scale 1 (train_step caches the value it set)
        for name, value in getattr(self, "_options", {}).items():       # per-model options survive a re-created handle
            N.check(N.lib().adm_unet_set_option(h, name.encode(), int(value)))
        return h

    def set_option(self, name: str, value: int):
        """Per-MODEL runtime option of the native executor (`adm_unet_set_option`, include/adm.h; not part of the reference's API).
        "wino6": this model's Winograd F(4x4) layer rule — 0 follows the process-wide option, 256 is the single-sample latency rule;
        "single_sample": 1 = the partition rules for a model sampled one spectrogram at a time (layers whose tiles cannot fill the chip with one
        sample split their input channels over more workgroups), 0 follows the process-wide option, -1 off. `AudioDiffusion` selects both for its own model."""
        if not hasattr(self, "_options"):
            self._options = {}
        if self._handle is not None:           # (a rejected value raises here and is not remembered)
            N.check(N.lib().adm_unet_set_option(self._handle, name.encode(), int(value)))
        self._options[name] = int(value)
        return self

    def _upload(self, key, t):
        t = t.detach().to(torch.float32).cpu().contiguous()
        N.check(N.lib().adm_unet_set_param(self._handle, key.encode(), C.c_void_p(t.data_ptr()), t.numel()))

    def _free(self):
        if self._handle is not None:
            N.lib().adm_unet_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass

    # ---- weights ---------------------------------------------------------------------------------------
    def load_state_dict(self, sd, strict=True):
        specs = {k: s for k, s, _ in param_specs(self.config)}
        new = {}
        for k, v in sd.items():
            ck = _canon_key(k)
            if ck not in specs:
                if strict:
                    raise KeyError(f"unexpected key {k}")
                continue
            v = v.reshape(specs[ck]) if tuple(v.shape) != tuple(specs[ck]) and v.numel() == math.prod(specs[ck]) else v
            if tuple(v.shape) != tuple(specs[ck]):
                raise ValueError(f"shape mismatch for {k}: {tuple(v.shape)} vs {specs[ck]}")
            new[ck] = v.detach().to(torch.float32).cpu().contiguous()
        missing = [k for k in specs if k not in new]
        if strict and missing:
            raise KeyError(f"missing keys: {missing[:8]}{'...' if len(missing) > 8 else ''}")
        self._sd.update(new)
        self._free()  # re-created (and re-uploaded) lazily at the next forward
        return self

    def state_dict(self):
        return dict(self._sd)

    def init_random(self, seed=0):
        """torch's default nn.Conv2d / nn.Linear / nn.GroupNorm initialisation (kaiming_uniform(a=sqrt(5)) ==
        U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias), seeded — for benchmarks and parity tests."""
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for k, shape, fan_in in param_specs(self.config):
            if fan_in == 0:
                sd[k] = torch.ones(shape)
            elif fan_in == -1:
                sd[k] = torch.zeros(shape)
            else:
                b = 1.0 / math.sqrt(fan_in)
                sd[k] = (torch.rand(shape, generator=g) * 2 - 1) * b
        return self.load_state_dict(sd)

    def num_parameters(self):
        return sum(math.prod(s) for _, s, _ in param_specs(self.config))

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    # ---- forward ---------------------------------------------------------------------------------------
    def _timesteps(self, timestep, B):
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t])
        t = t.detach().to("cpu", torch.float32).reshape(-1)
        if t.numel() not in (1, B):
            raise ValueError("timestep must be a scalar or have one entry per sample")
        return t.contiguous()

    def forward(self, sample, timestep, return_dict=True):
        assert sample.dim() == 4 and sample.dtype == torch.float32
        if tuple(sample.shape[2:]) != self._hw():
            self.sample_size = tuple(sample.shape[2:])
        h = self._ensure_handle()
        x = sample.contiguous()
        B = x.shape[0]
        t = self._timesteps(timestep, B)
        out = torch.empty((B, self.config.out_channels) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        N.check(N.lib().adm_unet_forward(h, N.ptr(x), C.cast(t.data_ptr(), N.c_float_p), t.numel(),
                                         N.ptr(out), B, N.stream_for(x)))
        return UNetOutput(sample=out)

    __call__ = forward

    # ---- training (scripts/train_unet.py:257-267) ---------------------------------------------------------------
    def enable_training(self, sample_hw=None, mixed_precision="no"):
        """Moves the master parameters into ONE flat device buffer (so the fused AdamW/EMA kernel updates them in a single
        launch) and switches the native executor to training mode (activations kept, gradient buffers, wgrad/dgrad
        weight packings). Returns (flat_params, flat_grads) — torch views the optimizer side works on.
        mixed_precision="bf16" (scripts/train_unet.py:391-401): the 3x3 stride-1 convolutions of the training passes take
        bf16 MFMA operands (the activated input rounded once per layer into a blocked 16-bit image, filters re-rounded from
        the fp32 masters after every optimizer step) with fp32 accumulation; everything the model keeps — weights,
        activations, gradients, optimizer state — stays fp32, so checkpoints and the fp32 sampling path are unaffected."""
        from .training import FlatBuffer
        if mixed_precision not in ("no", "bf16", "fp16"):
            raise ValueError(f"mixed_precision must be 'no', 'bf16' or 'fp16', got {mixed_precision!r}")
        # level 3 (default, round 4; measured 69 vs 84 ms per step at level 2, profiles/r04_*): level 2 plus the blocked
        # operand images of k_conv_bf16b.hip for the 3x3 stride-1 convolutions of all three passes.  Level 2 (round 2; 98.6
        # vs 113.5 ms at level 1): the 1x1 convolutions and the stride-2 data gradients take bf16 operands too;
        # ADM_BF16_LEVEL=1 keeps them fp32.
        # The level is read by adm_unet_enable_training and belongs to THIS model from then on: the process-wide option is
        # put back to 0 below, and the native sampling entry points always run fp32.
        # "fp16" (`train_unet.py:391-395`): the same kernels on IEEE binary16 operands (v_mfma_f32_32x32x16_f16); binary16's
        # narrow exponent needs the gradient side scaled: training.GradScaler + train_step(loss_scale=), as accelerate's
        # torch.cuda.amp.GradScaler does for the reference.
        level = int(os.environ.get("ADM_BF16_LEVEL", "3")) if mixed_precision in ("bf16", "fp16") else 0
        N.check(N.lib().adm_set_option(b"conv_bf16", level))
        N.check(N.lib().adm_set_option(b"conv_op16_f16", int(mixed_precision == "fp16")))
        self.mixed_precision = mixed_precision
        if sample_hw is not None:
            self.sample_size = tuple(sample_hw)
        self._free()
        dev = N.default_device()
        specs = [(k, s) for k, s, _ in param_specs(self.config)]
        self.flat = FlatBuffer(specs, dev).load(self._sd)
        self.flat_grads = torch.zeros_like(self.flat.data)
        h = self._create_handle()
        lib = N.lib()
        for k, _ in specs:
            N.check(lib.adm_unet_bind_param(h, k.encode(), C.c_void_p(self.flat.view(k).data_ptr())))
        try:
            N.check(lib.adm_unet_enable_training(h, N.ptr(self.flat.data), self.flat.numel))
        finally:
            N.check(lib.adm_set_option(b"conv_bf16", 0))
            N.check(lib.adm_set_option(b"conv_op16_f16", 0))
        self._training = True
        return self.flat.data, self.flat_grads

    def train_step(self, noisy, timesteps, target, loss_scale=1.0):
        """loss = mse(unet(noisy, t), target) and its gradient w.r.t. every parameter (into `flat_grads`). loss_scale (fp16):
        the gradients carry that factor (the returned loss does not); `training.GradScaler` un-scales them."""
        assert getattr(self, "_training", False), "call enable_training() first"
        if float(loss_scale) != self._loss_scale:      # _create_handle resets the cache together with the native value
            N.check(N.lib().adm_unet_set_loss_scale(self._handle, float(loss_scale)))
            self._loss_scale = float(loss_scale)
        x, tgt = noisy.contiguous(), target.contiguous()
        B = x.shape[0]
        t = self._timesteps(timesteps, B)
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        N.check(N.lib().adm_unet_forward_backward(self._handle, N.ptr(x), C.cast(t.data_ptr(), N.c_float_p), t.numel(),
                                                  N.ptr(tgt), N.ptr(loss), N.ptr(self.flat_grads), B, N.stream_for(x)))
        return loss[0]

    def refresh_weights(self):
        """Re-pack the derived weight layouts after the optimizer changed the flat master parameters."""
        N.check(N.lib().adm_unet_refresh_weights(self._handle, None))

    def sync_state_dict_from_flat(self):
        for k in list(self._sd.keys()):
            self._sd[k] = self.flat.view(k).detach().cpu().clone()

    # ---- diffusers on-disk layout ------------------------------------------------------------------------
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
        d = {"_class_name": type(self).__name__, "_diffusers_version": "0.24.0"}
        d.update({k: (list(v) if isinstance(v, tuple) else v) for k, v in self.config.items()})
        d["sample_size"] = list(self.sample_size) if isinstance(self.sample_size, tuple) else self.sample_size
        with open(os.path.join(path, self.config_name), "w") as f:
            json.dump(d, f, indent=2, sort_keys=True)
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(self._sd, os.path.join(path, "diffusion_pytorch_model.safetensors"))
        else:
            torch.save(self._sd, os.path.join(path, "diffusion_pytorch_model.bin"))


# ---------------------------------------------------------------------------------------------- conditional model
_COND_DEFAULTS = dict(
    sample_size=None, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True, freq_shift=0,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    mid_block_type="UNetMidBlock2DCrossAttn",
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    only_cross_attention=False, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1,
    mid_block_scale_factor=1, dropout=0.0, act_fn="silu", norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=1280,
    transformer_layers_per_block=1, encoder_hid_dim=None, encoder_hid_dim_type=None, attention_head_dim=8,
    num_attention_heads=None, dual_cross_attention=False, use_linear_projection=False, class_embed_type=None,
    addition_embed_type=None, num_class_embeds=None, upcast_attention=False, resnet_time_scale_shift="default",
    time_embedding_type="positional", time_embedding_dim=None, time_embedding_act_fn=None, timestep_post_act=None,
    time_cond_proj_dim=None, conv_in_kernel=3, conv_out_kernel=3, class_embeddings_concat=False,
)


class UNet2DConditionModel(UNet2DModel):
    """diffusers' UNet2DConditionModel as the reference builds it (`scripts/train_unet.py:139-159`): CrossAttn down / up
    blocks and a cross-attention mid block whose Transformer2DModel attends to `encoding` (batch, seq_length,
    cross_attention_dim) — `self.unet(images, t, encoding)["sample"]` (`pipeline_audio_diffusion.py:160-161`).
    `attention_head_dim` is the number of heads, as in diffusers 0.24 (`num_attention_heads or attention_head_dim`).
    Training: `enable_training()` + `train_step(noisy, t, target, encoding)`."""

    _defaults = _COND_DEFAULTS

    @staticmethod
    def _validate(cfg):
        ok_down, ok_up = {"DownBlock2D", "CrossAttnDownBlock2D"}, {"UpBlock2D", "CrossAttnUpBlock2D"}
        bad = [t for t in cfg["down_block_types"] if t not in ok_down] + [t for t in cfg["up_block_types"] if t not in ok_up]
        if bad:
            raise NotImplementedError(f"block types {bad} are not implemented (Down/CrossAttnDown/Up/CrossAttnUp)")
        for k, want in (("time_embedding_type", "positional"), ("act_fn", "silu"), ("resnet_time_scale_shift", "default"),
                        ("center_input_sample", False), ("class_embed_type", None), ("downsample_padding", 1),
                        ("mid_block_scale_factor", 1), ("mid_block_type", "UNetMidBlock2DCrossAttn"),
                        ("only_cross_attention", False), ("transformer_layers_per_block", 1), ("encoder_hid_dim", None),
                        ("num_attention_heads", None), ("dual_cross_attention", False), ("use_linear_projection", False),
                        ("addition_embed_type", None), ("upcast_attention", False), ("time_embedding_dim", None),
                        ("time_cond_proj_dim", None), ("conv_in_kernel", 3), ("conv_out_kernel", 3)):
            if cfg.get(k, want) != want:
                raise NotImplementedError(f"UNet2DConditionModel config {k}={cfg[k]!r} is not implemented (expected {want!r})")
        if not isinstance(cfg["attention_head_dim"], int) or not isinstance(cfg["cross_attention_dim"], int):
            raise NotImplementedError("per-block attention_head_dim / cross_attention_dim tuples are not implemented")
        if len(cfg["block_out_channels"]) > 8:
            raise NotImplementedError("more than 8 blocks")

    def forward(self, sample, timestep, encoder_hidden_states, return_dict=True):
        assert sample.dim() == 4 and sample.dtype == torch.float32
        if tuple(sample.shape[2:]) != self._hw():
            self.sample_size = tuple(sample.shape[2:])
        self._set_encoding(self._ensure_handle(), encoder_hidden_states, sample.shape[0], sample.device)
        return UNet2DModel.forward(self, sample, timestep)

    def _set_encoding(self, h, enc, B, device):
        if enc is None:
            raise ValueError("UNet2DConditionModel needs `encoding` (batch, seq_length, cross_attention_dim)")
        enc = enc.to(device, torch.float32)
        if enc.dim() == 2:
            enc = enc[:, None, :]
        if enc.dim() != 3 or enc.shape[0] != B or enc.shape[2] != self.config.cross_attention_dim:
            raise ValueError(f"encoding shape {tuple(enc.shape)} does not match (batch={B}, seq, {self.config.cross_attention_dim})")
        self._enc = enc.contiguous()            # kept alive: the native side stores the pointer
        N.check(N.lib().adm_unet_set_encoding(h, N.ptr(self._enc), self._enc.shape[1]))

    __call__ = forward

    def train_step(self, noisy, timesteps, target, encoder_hidden_states, loss_scale=1.0):
        """`model(noisy_images, timesteps, batch["encoding"])` + mse + backward (scripts/train_unet.py:254-259)."""
        assert getattr(self, "_training", False), "call enable_training() first"
        self._set_encoding(self._handle, encoder_hidden_states, noisy.shape[0], noisy.device)
        return UNet2DModel.train_step(self, noisy, timesteps, target, loss_scale=loss_scale)
