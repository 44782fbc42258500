"""AudioDiffusionPipeline drop-in (`audiodiffusion/pipeline_audio_diffusion.py:39-258` of the reference).

Same constructor, `__call__` signature/defaults (`:72-87`), `encode` (`:208`), `slerp` (`:244`),
`get_default_steps` (`:63`) and return types. The denoising loop (`:159-185`), the scheduler step, the mask
overwrite and the uint8 dequantisation (`:192-194`) run as ONE native call (`adm_sample_loop`: a captured
hipGraph of {UNet forward, fused scheduler epilogue} replayed per step, csrc/unet_exec.hip); the audio codec
runs in the Mel HIP kernels. `diffusers` is not required: a minimal `DiffusionPipeline`-compatible base
(`from_pretrained / save_pretrained / to / device / progress_bar / register_modules`) reads and writes the
diffusers on-disk layout (`model_index.json`, `unet/`, `scheduler/`, `mel/`).

Kept VERBATIM from the reference (GPL-3.0, see NOTICE.md at the repository root), because a drop-in must keep their quirks
bit for bit and `tests/test_reference_pin.py` pins them against the reference's own program text: the `__call__` signature
(`:72-87`), the audio-conditioned prologue (`:134-156`: slice -> image -> [-1, 1] tensor, the `images[0, 0] = add_noise(...)`
write-through into the aliased `noise`, the `(B, steps, H, W)` mask) and the image -> tensor lines of `encode` (`:221-226`).
Everything else in this file — the base class, the native loop call, dequantisation, batched image -> audio — is original.
"""
import ctypes as C
import json
import os
from math import acos, sin
from typing import List, Tuple, Union

import numpy as np
import torch
from PIL import Image

from . import _native as N
from . import ops
from .mel import Mel
from .schedulers import DDIMScheduler, DDPMScheduler, randn_tensor
from .unet import UNet2DConditionModel, UNet2DModel
from .vae import AutoencoderKL


class PipelineOutput(dict):
    """BaseOutput-like: attribute and key access to `audios` / `images`."""
    __getattr__ = dict.__getitem__


_CLASSES = {"AutoencoderKL": AutoencoderKL, "UNet2DModel": UNet2DModel, "UNet2DConditionModel": UNet2DConditionModel,
            "DDIMScheduler": DDIMScheduler, "DDPMScheduler": DDPMScheduler, "Mel": Mel}


class DiffusionPipeline:
    """The subset of diffusers.DiffusionPipeline the reference relies on (`__init__.py:30-32`, `train_unet.py:303`)."""
    config_name = "model_index.json"
    _optional_components: List[str] = []

    def __init__(self):
        self._modules = {}
        self._progress_bar_config = {}
        self._device = N.default_device()

    def register_modules(self, **kwargs):
        for k, v in kwargs.items():
            self._modules[k] = v
            setattr(self, k, v)

    @property
    def device(self):
        return self._device

    def to(self, device=None, *a, **k):
        if device is not None:
            d = torch.device(device)
            if d.type == "cuda" and d.index is None:
                d = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
            if N.is_device_build() and d.type != "cuda":
                raise RuntimeError("this pipeline runs on the MI355X only (HIP kernels, no CPU path)")
            self._device = d
        return self

    def set_progress_bar_config(self, **kwargs):
        self._progress_bar_config = kwargs

    def progress_bar(self, iterable=None, total=None):
        cfg = dict(self._progress_bar_config)
        if cfg.get("disable"):
            return iterable
        try:
            from tqdm.auto import tqdm
            return tqdm(iterable, total=total, **cfg)
        except Exception:
            return iterable

    @classmethod
    def from_pretrained(cls, path, **kwargs):
        if not os.path.isdir(path):
            raise EnvironmentError(f"{path} is not a local directory (there is no hub access; pass a diffusers-layout directory)")
        with open(os.path.join(path, cls.config_name)) as f:
            index = json.load(f)
        comps = {}
        for name, spec in index.items():
            if name.startswith("_"):
                continue
            lib, klass = spec
            if klass is None:
                comps[name] = None
                continue
            if klass not in _CLASSES:
                raise NotImplementedError(f"component {name}: class {klass} is not implemented on this path")
            comps[name] = _CLASSES[klass].from_pretrained(path, subfolder=name)
        for opt in cls._optional_components:
            comps.setdefault(opt, None)
        return cls(**comps)

    def save_pretrained(self, path, safe_serialization=True):
        os.makedirs(path, exist_ok=True)
        index = {"_class_name": type(self).__name__, "_diffusers_version": "0.24.0"}
        for name, m in self._modules.items():
            if m is None:
                index[name] = [None, None]
                continue
            lib = "audio_diffusion" if isinstance(m, Mel) else "diffusers"
            index[name] = [lib, type(m).__name__]
            sub = os.path.join(path, name)
            if isinstance(m, (UNet2DModel, AutoencoderKL)):
                m.save_pretrained(sub, safe_serialization=safe_serialization)
            else:
                m.save_pretrained(sub)
        with open(os.path.join(path, self.config_name), "w") as f:
            json.dump(index, f, indent=2, sort_keys=True)


class AudioDiffusionPipeline(DiffusionPipeline):
    """
    Parameters (as the reference, `pipeline_audio_diffusion.py:39-61`):
        vqvae: AutoencoderKL for latent audio diffusion or None
        unet: UNet2DModel
        mel: Mel — transform audio <-> spectrogram
        scheduler: DDIMScheduler or DDPMScheduler
    """

    _optional_components = ["vqvae"]
    # per-call step-noise staging is bounded: the loop runs in chunks of this many steps
    _STEP_CHUNK = 100

    def __init__(self, vqvae, unet, mel, scheduler):
        super().__init__()
        self.register_modules(unet=unet, scheduler=scheduler, mel=mel, vqvae=vqvae)

    def get_default_steps(self) -> int:
        return 50 if isinstance(self.scheduler, DDIMScheduler) else 1000

    # ---- the native denoising loop (pipeline_audio_diffusion.py:159-185 + :192-194) ----------------------
    def _denoise(self, images, start_step, eta, step_generator, mask, mask_start, mask_end, step_noise=None,
                 use_graph=True, want_u8=True, encoding=None, stop_step=None):
        """The denoising loop (`:159-185`) as ONE native call per chunk of steps. `stop_step` (tests only) ends the loop
        before that step index, so that a single step of a long schedule can be compared in isolation."""
        sched, unet = self.scheduler, self.unet
        rows = sched.coef_rows(eta)[start_step:stop_step]
        n = len(rows)
        x = images.contiguous().clone()  # the reference never writes the loop state back into `noise`
        B, Cc, H, W = x.shape
        if tuple(unet._hw()) != (H, W):
            unet.sample_size = (H, W)
        h = unet._ensure_handle()
        if isinstance(unet, UNet2DConditionModel):     # `self.unet(images, t, encoding)` (:160-161): constant over the loop
            unet._set_encoding(h, encoding, B, x.device)
        u8 = torch.empty((B, H, W, Cc), dtype=torch.uint8, device=x.device) if want_u8 else None
        if Cc != 1 and want_u8:
            u8 = None  # NHWC permute for multi-channel images is done after the loop
        needs_noise = [r["k_noise"] != 0.0 for r in rows]
        chunk = n if not any(needs_noise) else min(n, self._STEP_CHUNK)
        stage = None
        done = 0
        while done < n:
            m = min(chunk, n - done)
            sub = rows[done:done + m]
            coef = (N.SchedCoef * m)(*[N.SchedCoef(*[float(r[k]) for k in
                                       ("sqrt_beta", "sqrt_alpha", "clip", "k_x0", "k_x", "k_eps", "k_noise", "timestep")])
                                       for r in sub])
            noise_ptr = None
            if any(needs_noise[done:done + m]):
                if stage is None or stage.shape[0] != m:
                    stage = torch.empty((m, B, Cc, H, W), dtype=torch.float32, device=x.device)
                for i in range(m):
                    if needs_noise[done + i]:
                        if step_noise is not None:
                            stage[i].copy_(step_noise[done + i])
                        else:  # same draw order as scheduler.step's randn_tensor calls in the reference loop
                            stage[i].copy_(randn_tensor((B, Cc, H, W), step_generator, x.device, torch.float32))
                noise_ptr = N.ptr(stage)
            last = done + m == n
            mask_ptr = None
            if mask is not None:
                # (B, n_total, H, W): this chunk starts at row `done`; the kernel indexes mask[:, step_in_chunk]
                mask_chunk = mask[:, done:done + m].contiguous()
                mask_ptr = N.ptr(mask_chunk)
            N.check(N.lib().adm_sample_loop(h, N.ptr(x), B, coef, m, noise_ptr, mask_ptr, int(mask_start), int(mask_end),
                                            N.ptr(u8) if (last and u8 is not None) else None, int(use_graph),
                                            N.stream_for(x)))
            if x.is_cuda and (noise_ptr is not None or mask_ptr is not None) and not last:
                torch.cuda.current_stream(x.device).synchronize()  # staging buffers are rewritten next chunk
            done += m
        return x, u8

    @torch.no_grad()
    def __call__(
        self,
        batch_size: int = 1,
        audio_file: str = None,
        raw_audio: np.ndarray = None,
        slice: int = 0,
        start_step: int = 0,
        steps: int = None,
        generator: torch.Generator = None,
        mask_start_secs: float = 0,
        mask_end_secs: float = 0,
        step_generator: torch.Generator = None,
        eta: float = 0,
        noise: torch.Tensor = None,
        encoding: torch.Tensor = None,
        return_dict=True,
        step_noise=None,
        audio=True,
        return_float=False,
        init_phase=None,
    ) -> Union[PipelineOutput, Tuple[List[Image.Image], Tuple[int, List[np.ndarray]]]]:
        """Generate random mel spectrogram from audio input and convert to audio (reference docstring `:89-112`).

        Extra keyword-only knobs (not in the reference; defaults reproduce it): `step_noise` injects the
        per-step scheduler noise (parity tests), `audio=False` skips the image->audio conversion,
        `return_float=True` additionally returns the final float images, `init_phase` (B, n_bins, frames) replaces the
        unseeded Griffin-Lim start phase librosa draws (`mel.py:165-167`)."""
        steps = steps or self.get_default_steps()
        self.scheduler.set_timesteps(steps)
        step_generator = step_generator or generator
        # For backwards compatibility
        if type(self.unet.sample_size) == int:
            self.unet.sample_size = (self.unet.sample_size, self.unet.sample_size)
        if noise is None:
            # the reference's torch.randn(..., generator=generator, device=self.device) (:120-128); going through
            # randn_tensor additionally accepts a CPU generator on the GPU build (drawn on the host, then moved)
            noise = randn_tensor(
                (batch_size, self.unet.in_channels, self.unet.sample_size[0], self.unet.sample_size[1]),
                generator, self.device, torch.float32)
        images = noise
        mask = None
        mask_start = mask_end = 0

        if audio_file is not None or raw_audio is not None:
            self.mel.load_audio(audio_file, raw_audio)
            input_image = self.mel.audio_slice_to_image(slice)
            input_image = np.frombuffer(input_image.tobytes(), dtype="uint8").reshape(
                (input_image.height, input_image.width)
            )
            input_image = (input_image / 255) * 2 - 1
            input_images = torch.tensor(input_image[np.newaxis, :, :], dtype=torch.float).to(self.device)

            if self.vqvae is not None:
                input_images = self.vqvae.encode(torch.unsqueeze(input_images, 0)).latent_dist.sample(
                    generator=generator
                )[0]
                input_images = 0.18215 * input_images

            if start_step > 0:
                images[0, 0] = self.scheduler.add_noise(input_images, noise, self.scheduler.timesteps[start_step - 1])

            pixels_per_second = (
                self.unet.sample_size[1] * self.mel.get_sample_rate() / self.mel.x_res / self.mel.hop_length
            )
            mask_start = int(mask_start_secs * pixels_per_second)
            mask_end = int(mask_end_secs * pixels_per_second)
            mask = self.scheduler.add_noise(input_images, noise, self.scheduler.timesteps[start_step:].clone())

        use_mask = mask if (mask is not None and (mask_start > 0 or mask_end > 0)) else None
        images, u8 = self._denoise(images, start_step, eta, step_generator, use_mask, mask_start, mask_end,
                                   step_noise=step_noise, want_u8=self.vqvae is None, encoding=encoding)

        if self.vqvae is not None:
            # 0.18215 was scaling factor used in training to ensure unit variance (pipeline:187-190); the 1/0.18215
            # multiply is folded into the decoder launch
            images = self.vqvae.decode(images, _in_scale=1 / 0.18215)["sample"]
            u8 = ops.dequant_u8(images).permute(0, 2, 3, 1).contiguous()
        final_float = images

        if u8 is None:  # multi-channel: dequantise then NHWC (pipeline:192-194)
            u8 = ops.dequant_u8(images).permute(0, 2, 3, 1).contiguous()
        arr = u8.cpu().numpy()
        images = list(
            map(lambda _: Image.fromarray(_[:, :, 0]), arr)
            if arr.shape[3] == 1
            else map(lambda _: Image.fromarray(_, mode="RGB").convert("L"), arr)
        )

        # the reference maps image_to_audio over the images one by one on a host core (:201); here the whole batch goes
        # through the Mel kernels in one launch sequence
        audios = list(self.mel.images_to_audios(images, init_phase=init_phase)) if audio else []
        if return_float:
            return images, final_float
        if not return_dict:
            return images, (self.mel.get_sample_rate(), audios)
        return PipelineOutput(audios=np.array(audios)[:, np.newaxis, :] if audio else np.zeros((0, 1, 0)), images=images)

    @torch.no_grad()
    def encode(self, images: List[Image.Image], steps: int = 50) -> torch.Tensor:
        """Reverse step process: recover noisy image from generated image (`pipeline_audio_diffusion.py:207-242`)."""
        # Only works with DDIM as this method is deterministic
        assert isinstance(self.scheduler, DDIMScheduler)
        if isinstance(self.unet, UNet2DConditionModel):   # the reference calls self.unet(sample, t) here (:237): no encoding
            raise NotImplementedError("encode() is defined for the unconditional UNet2DModel only, as in the reference")
        self.scheduler.set_timesteps(steps)
        sample = np.array(
            [np.frombuffer(image.tobytes(), dtype="uint8").reshape((1, image.height, image.width)) for image in images]
        )
        sample = (sample / 255) * 2 - 1
        sample = torch.Tensor(sample).to(self.device).contiguous()
        rows = self.scheduler.encode_rows()
        coef = (N.SchedCoef * len(rows))(*[N.SchedCoef(*[float(r[k]) for k in
                                           ("sqrt_beta", "sqrt_alpha", "clip", "k_x0", "k_x", "k_eps", "k_noise", "timestep")])
                                           for r in rows])
        B, _, H, W = sample.shape
        if tuple(self.unet._hw()) != (H, W):
            self.unet.sample_size = (H, W)
        h = self.unet._ensure_handle()
        N.check(N.lib().adm_encode_loop(h, N.ptr(sample), B, coef, len(rows), 1, N.stream_for(sample)))
        return sample

    @staticmethod
    def slerp(x0: torch.Tensor, x1: torch.Tensor, alpha: float) -> torch.Tensor:
        """Spherical Linear intERPolation (`pipeline_audio_diffusion.py:244-258`)."""
        theta = acos(torch.dot(torch.flatten(x0), torch.flatten(x1)) / torch.norm(x0) / torch.norm(x1))
        return sin((1 - alpha) * theta) * x0 / sin(theta) + sin(alpha * theta) * x1 / sin(theta)
