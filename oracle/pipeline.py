"""Oracle of the sampling procedure `audiodiffusion/pipeline_audio_diffusion.py:39-258` (TEST INFRASTRUCTURE ONLY).

Restates `AudioDiffusionPipeline.__call__` (:71-205), `encode` (:207-242) and `slerp` (:244-258) on torch-CPU with the oracle
UNet / schedulers / Mel, decomposed into the stages the HIP path fuses (start state, audio conditioning, one denoising
step, image conversion).  Behaviour the parity tests depend on is kept exactly: the loop state IS the caller's `noise`
tensor (:131), only element [0, 0] receives the noised input when `start_step > 0` (:150) and the mask is built from the
noise AFTER that write (:157), the mask overwrite indexes `mask[:, step]` (:181-185), the uint8 conversion is numpy's
round-half-to-even (:194).  Extensions for parity tests only: `step_noise` (per-step noise tensors instead of the
scheduler's own `randn_tensor` draws), `audio=False` (skip the serial image_to_audio map, :201), `return_float`,
`init_phase` (Griffin-Lim start phases instead of librosa's unseeded draw).
"""
from math import acos, sin

import numpy as np
import torch
from PIL import Image

from .schedulers import DDIMScheduler

LATENT_SCALE = 0.18215          # :147,189 (hard-coded in the reference, not config.scaling_factor)


class AudioDiffusionPipeline:
    def __init__(self, vqvae, unet, mel, scheduler):
        self.vqvae, self.unet, self.mel, self.scheduler = vqvae, unet, mel, scheduler
        self.device = torch.device("cpu")

    def progress_bar(self, it):
        return it

    def get_default_steps(self):                                   # :63-69
        return 50 if isinstance(self.scheduler, DDIMScheduler) else 1000

    # ---- stages ---------------------------------------------------------------------------------------------
    def _sample_hw(self):                                          # :118-119 (int sample_size kept for old checkpoints)
        ss = self.unet.sample_size
        if type(ss) == int:
            self.unet.sample_size = ss = (ss, ss)
        return ss

    def _input_as_model_space(self, audio_file, raw_audio, slice, generator):
        """:135-147 — one slice of the input as a [-1, 1] image (or its scaled VAE latent), shape (C, H, W)."""
        self.mel.load_audio(audio_file, raw_audio)
        img = self.mel.audio_slice_to_image(slice)
        px = np.frombuffer(img.tobytes(), dtype="uint8").reshape((img.height, img.width))
        x = torch.tensor(((px / 255) * 2 - 1)[np.newaxis, :, :], dtype=torch.float)
        if self.vqvae is not None:
            x = LATENT_SCALE * self.vqvae.encode(x.unsqueeze(0)).latent_dist.sample(generator=generator)[0]
        return x

    def _predict(self, images, t, encoding):                       # :160-163
        conditional = hasattr(self.unet.config, "get") and self.unet.config.get("cross_attention_dim")
        out = self.unet(images, t, encoding) if conditional else self.unet(images, t)
        return out["sample"]

    def _advance(self, eps, t, images, eta, step_generator, variance_noise):   # :165-179
        kw = dict(model_output=eps, timestep=t, sample=images, generator=step_generator, variance_noise=variance_noise)
        if isinstance(self.scheduler, DDIMScheduler):
            kw["eta"] = eta
        return self.scheduler.step(**kw)["prev_sample"]

    def _to_pil(self, images):                                     # :192-199
        arr = (images / 2 + 0.5).clamp(0, 1).cpu().permute(0, 2, 3, 1).numpy()
        arr = (arr * 255).round().astype("uint8")
        if arr.shape[3] == 1:
            return [Image.fromarray(a[:, :, 0]) for a in arr]
        return [Image.fromarray(a, mode="RGB").convert("L") for a in arr]

    # ---- the procedure ----------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, batch_size=1, audio_file=None, raw_audio=None, slice=0, start_step=0, steps=None,
                 generator=None, mask_start_secs=0, mask_end_secs=0, step_generator=None, eta=0, noise=None,
                 encoding=None, return_dict=True, step_noise=None, audio=True, return_float=False, init_phase=None):
        sched = self.scheduler
        sched.set_timesteps(steps or self.get_default_steps())
        step_generator = step_generator or generator
        hw = self._sample_hw()
        if noise is None:
            noise = torch.randn((batch_size, self.unet.in_channels, hw[0], hw[1]), generator=generator)
        images = noise                                             # aliasing on purpose (:131)
        mask, mask_start, mask_end = None, 0, 0
        if audio_file is not None or raw_audio is not None:
            x_in = self._input_as_model_space(audio_file, raw_audio, slice, generator)
            if start_step > 0:                                     # :149-150 — writes through the alias into `noise`
                images[0, 0] = sched.add_noise(x_in, noise, sched.timesteps[start_step - 1])
            px_per_sec = hw[1] * self.mel.get_sample_rate() / self.mel.x_res / self.mel.hop_length
            mask_start, mask_end = int(mask_start_secs * px_per_sec), int(mask_end_secs * px_per_sec)
            mask = sched.add_noise(x_in, noise, sched.timesteps[start_step:].clone())   # (B, n_steps, H, W), :157
        for step, t in enumerate(self.progress_bar(sched.timesteps[start_step:])):
            eps = self._predict(images, t, encoding)
            images = self._advance(eps, t, images, eta, step_generator, None if step_noise is None else step_noise[step])
            if mask is not None and mask_start > 0:
                images[:, :, :, :mask_start] = mask[:, step, :, :mask_start]
            if mask is not None and mask_end > 0:
                images[:, :, :, -mask_end:] = mask[:, step, :, -mask_end:]
        if self.vqvae is not None:                                 # :187-190
            images = self.vqvae.decode(1 / LATENT_SCALE * images)["sample"]
        final_float = images
        pil = self._to_pil(images)
        audios = [self.mel.image_to_audio(im, init_phase=None if init_phase is None else np.asarray(init_phase)[i])
                  for i, im in enumerate(pil)] if audio else []
        if return_float:
            return pil, final_float
        if not return_dict:
            return pil, (self.mel.get_sample_rate(), audios)
        return dict(audios=np.array(audios)[:, np.newaxis, :], images=pil)

    @torch.no_grad()
    def encode(self, images, steps=50):
        """DDIM inversion (:207-242): walk the timesteps upwards, re-noising with the model's own prediction."""
        sched = self.scheduler
        assert isinstance(sched, DDIMScheduler)
        sched.set_timesteps(steps)
        px = np.array([np.frombuffer(im.tobytes(), dtype="uint8").reshape((1, im.height, im.width)) for im in images])
        sample = torch.Tensor((px / 255) * 2 - 1)
        stride = sched.config.num_train_timesteps // sched.num_inference_steps
        for t in self.progress_bar(torch.flip(sched.timesteps, (0,))):
            a_t = sched.alphas_cumprod[t]
            a_prev = sched.alphas_cumprod[t - stride] if t - stride >= 0 else sched.final_alpha_cumprod
            eps = self.unet(sample, t)["sample"]
            sample = (sample - (1 - a_prev) ** (0.5) * eps) * a_prev ** (-0.5)      # predicted x0 ...
            sample = sample * a_t ** (0.5) + (1 - a_t) ** (0.5) * eps               # ... re-noised to level t
        return sample

    @staticmethod
    def slerp(x0, x1, alpha):                                      # :244-258
        a, b = torch.flatten(x0), torch.flatten(x1)
        theta = acos(torch.dot(a, b) / torch.norm(x0) / torch.norm(x1))
        return (sin((1 - alpha) * theta) * x0 + sin(alpha * theta) * x1) / sin(theta)
