"""AudioEncoder drop-in (reference: `audiodiffusion/audio_encoder.py:62-107`): mel slices of a track -> 100-d encoding for
the conditional UNet (`scripts/train_unet.py:93-94,158`; `pipeline(..., encoding=...)`).

Same state-dict keys and on-disk layout (`config.json` + `diffusion_pytorch_model.{safetensors,bin}`, ModelMixin) as the
reference, same `encode(audio_files, pool)` contract.  Inference only, as the reference uses it (`encode` runs eval /
no_grad): Dropout is the identity and BatchNorm's running statistics are folded into one scale/shift per channel.  All
slices of a file go through the batched HIP Mel kernels and the encoder in one batch instead of a Python loop per slice.
"""
import ctypes as C
import json
import os

import numpy as np
import torch

from . import _native as N
from .mel import Mel

_CH = (1, 32, 64, 128)
_BN_EPS = 1e-3


class AudioEncoder:
    config_name = "config.json"

    def __init__(self):
        # audio_encoder.py:65-72
        self.mel = Mel(x_res=216, y_res=96, sample_rate=22050, n_fft=2048, hop_length=512, top_db=80)
        self._sd = {}
        self._dev = None
        self.device = N.default_device()

    # ---- weights -----------------------------------------------------------------------------------------
    def load_state_dict(self, sd):
        need = [f"conv_blocks.{i}.{k}" for i in range(3) for k in
                ("sep_conv.depthwise.weight", "sep_conv.pointwise.weight", "sep_conv.pointwise.bias", "batch_norm.weight",
                 "batch_norm.bias", "batch_norm.running_mean", "batch_norm.running_var")]
        need += ["dense_block.dense.weight", "dense_block.dense.bias", "dense_block.batch_norm.weight",
                 "dense_block.batch_norm.bias", "dense_block.batch_norm.running_mean", "dense_block.batch_norm.running_var",
                 "embedding.weight", "embedding.bias"]
        missing = [k for k in need if k not in sd]
        if missing:
            raise KeyError(f"missing keys: {missing[:6]}{'...' if len(missing) > 6 else ''}")
        self._sd = {k: v.detach().cpu().clone() for k, v in sd.items()}
        self._dev = None
        return self

    def state_dict(self):
        return dict(self._sd)

    def _weights(self):
        if self._dev is not None:
            return self._dev
        sd, d = self._sd, {}

        def up(t):
            return t.to(torch.float32).contiguous().to(self.device)

        def fold(p):        # BatchNorm in eval mode: y = (x - mean) / sqrt(var + eps) * w + b = x * scale + shift
            scale = sd[p + "weight"].double() / torch.sqrt(sd[p + "running_var"].double() + _BN_EPS)
            return up(scale), up(sd[p + "bias"].double() - sd[p + "running_mean"].double() * scale)

        for i in range(3):
            p = f"conv_blocks.{i}."
            d[f"dw{i}"] = up(sd[p + "sep_conv.depthwise.weight"])
            d[f"pw{i}"] = up(sd[p + "sep_conv.pointwise.weight"].reshape(_CH[i + 1], _CH[i]))
            d[f"pb{i}"] = up(sd[p + "sep_conv.pointwise.bias"])
            d[f"s{i}"], d[f"t{i}"] = fold(p + "batch_norm.")
        d["dw"], d["db"] = up(sd["dense_block.dense.weight"]), up(sd["dense_block.dense.bias"])
        d["ds"], d["dt"] = fold("dense_block.batch_norm.")
        d["ew"], d["eb"] = up(sd["embedding.weight"]), up(sd["embedding.bias"])
        self._dev = d
        return d

    # ---- forward (audio_encoder.py:78-84) ------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x):
        """x (n, 1, y_res, x_res) float in [0, 1] -> (n, 100)."""
        w = self._weights()
        x = x.to(self.device, torch.float32).contiguous()
        n, _, H, W = x.shape
        st = N.stream_for(x)
        lib = N.lib()
        for i in range(3):
            ci, co = _CH[i], _CH[i + 1]
            tmp = torch.empty((n, ci, H, W), dtype=torch.float32, device=x.device)
            y = torch.empty((n, co, H // 2, W // 2), dtype=torch.float32, device=x.device)
            N.check(lib.adm_sepconv_block(N.ptr(x), N.ptr(w[f"dw{i}"]), N.ptr(w[f"pw{i}"]), N.ptr(w[f"pb{i}"]),
                                          N.ptr(w[f"s{i}"]), N.ptr(w[f"t{i}"]), 0.2, N.ptr(tmp), N.ptr(y), n, ci, co, H, W, st))
            x, H, W = y, H // 2, W // 2
        K = _CH[3] * H * W
        if w["dw"].shape[1] != K:
            raise ValueError(f"dense_block expects {w['dw'].shape[1]} features, the input gives {K} (mel resolution mismatch)")
        h = torch.empty((n, w["dw"].shape[0]), dtype=torch.float32, device=x.device)
        N.check(lib.adm_dense_act(N.ptr(x), N.ptr(w["dw"]), N.ptr(w["db"]), N.ptr(w["ds"]), N.ptr(w["dt"]), 0.2, 1,
                                  N.ptr(h), n, K, h.shape[1], _CH[3], st))
        out = torch.empty((n, w["ew"].shape[0]), dtype=torch.float32, device=x.device)
        N.check(lib.adm_dense_act(N.ptr(h), N.ptr(w["ew"]), N.ptr(w["eb"]), None, None, 0.0, 0, N.ptr(out), n, h.shape[1],
                                  out.shape[1], 0, st))
        return out

    __call__ = forward

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    # ---- encode (audio_encoder.py:86-107) ------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, audio_files, pool="average"):
        y = []
        for audio_file in audio_files:
            self.mel.load_audio(audio_file)
            n = self.mel.get_number_of_slices()
            images = self.mel.audio_slices_to_images([self.mel.get_audio_slice(i) for i in range(n)])   # (n, y_res, x_res) u8
            x = torch.from_numpy(np.ascontiguousarray(images)).to(torch.float32)[:, None] / 255
            e = self(x)
            if pool == "average":
                e = torch.mean(e, dim=0)
            elif pool == "max":
                # the reference assigns torch.max(..., dim=0) — a (values, indices) pair — and then fails in torch.stack;
                # the values are what the pooling means
                e = torch.max(e, dim=0).values
            else:
                assert pool is None, f"Unknown pooling method {pool}"
            y += [e]
        return torch.stack(y)

    # ---- ModelMixin on-disk layout -----------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, path, subfolder=None):
        p = os.path.join(path, subfolder) if subfolder else path
        m = cls()
        st = os.path.join(p, "diffusion_pytorch_model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(p, "diffusion_pytorch_model.bin"), map_location="cpu", weights_only=True)
        return m.load_state_dict(sd)

    def save_pretrained(self, path, safe_serialization=True):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, self.config_name), "w") as f:
            json.dump({"_class_name": "AudioEncoder", "_diffusers_version": "0.24.0"}, f, indent=2, sort_keys=True)
        sd = {k: v.contiguous() for k, v in self._sd.items()}
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(path, "diffusion_pytorch_model.safetensors"))
        else:
            torch.save(sd, os.path.join(path, "diffusion_pytorch_model.bin"))
