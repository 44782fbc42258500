"""Oracle restatement of diffusers==0.24.0 DDPMScheduler / DDIMScheduler (TEST INFRASTRUCTURE ONLY).

Constructed by the reference with only `num_train_timesteps`
(`scripts/train_unet.py:161-164`) and used at
`audiodiffusion/pipeline_audio_diffusion.py:115,150,157,166-179,221-234`.
Restated in the same 0-d fp32 torch-tensor arithmetic diffusers uses
(SURVEY.md §8(a) rows S1-S5). Parity unpinned (see oracle/__init__.py).
"""
import math

import numpy as np
import torch


class _Config(dict):
    __getattr__ = dict.__getitem__


def _betas(num_train_timesteps, beta_start, beta_end, beta_schedule):
    if beta_schedule == "linear":
        return torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    if beta_schedule == "scaled_linear":
        return torch.linspace(beta_start**0.5, beta_end**0.5, num_train_timesteps, dtype=torch.float32) ** 2
    if beta_schedule == "squaredcos_cap_v2":
        f = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2  # noqa: E731
        b = [min(1 - f((i + 1) / num_train_timesteps) / f(i / num_train_timesteps), 0.999) for i in range(num_train_timesteps)]
        return torch.tensor(b, dtype=torch.float32)
    raise NotImplementedError(beta_schedule)


class _SchedulerBase:
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 clip_sample=True, clip_sample_range=1.0, prediction_type="epsilon",
                 timestep_spacing="leading", steps_offset=0, **extra):
        assert prediction_type == "epsilon" and timestep_spacing == "leading"
        self.config = _Config(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                              beta_schedule=beta_schedule, clip_sample=clip_sample,
                              clip_sample_range=clip_sample_range, prediction_type=prediction_type,
                              timestep_spacing=timestep_spacing, steps_offset=steps_offset, **extra)
        self.betas = _betas(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())

    def set_timesteps(self, num_inference_steps):  # row S1
        self.num_inference_steps = num_inference_steps
        step_ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
        ts += self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def add_noise(self, original_samples, noise, timesteps):  # row S4
        ac = self.alphas_cumprod.to(dtype=original_samples.dtype)
        sa = ac[timesteps] ** 0.5
        sa = sa.flatten()
        while len(sa.shape) < len(original_samples.shape):
            sa = sa.unsqueeze(-1)
        sb = (1 - ac[timesteps]) ** 0.5
        sb = sb.flatten()
        while len(sb.shape) < len(original_samples.shape):
            sb = sb.unsqueeze(-1)
        return sa * original_samples + sb * noise


class DDPMScheduler(_SchedulerBase):
    """variance_type fixed_small (row S3)."""

    def __init__(self, **kw):
        kw.setdefault("variance_type", "fixed_small")
        super().__init__(**kw)

    def step(self, model_output, timestep, sample, generator=None, variance_noise=None):
        t = int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        x0 = (sample - b_t ** (0.5) * model_output) / a_t ** (0.5)
        if self.config.clip_sample:
            x0 = x0.clamp(-self.config.clip_sample_range, self.config.clip_sample_range)
        c0 = (a_prev ** (0.5) * cur_b) / b_t
        c1 = cur_a ** (0.5) * b_prev / b_t
        prev = c0 * x0 + c1 * sample
        variance = 0
        if t > 0:
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype)
            var = (1 - a_prev) / (1 - a_t) * cur_b
            var = torch.clamp(var, min=1e-20)
            variance = (var ** 0.5) * variance_noise
        prev = prev + variance
        return {"prev_sample": prev, "pred_original_sample": x0}


class DDIMScheduler(_SchedulerBase):
    """set_alpha_to_one=True, use_clipped_model_output=False (row S2)."""

    def __init__(self, set_alpha_to_one=True, **kw):
        super().__init__(**kw)
        self.config["set_alpha_to_one"] = set_alpha_to_one
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]

    def step(self, model_output, timestep, sample, eta=0.0, generator=None, variance_noise=None):
        t = int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        x0 = (sample - b_t ** (0.5) * model_output) / a_t ** (0.5)
        eps = model_output
        if self.config.clip_sample:
            x0 = x0.clamp(-self.config.clip_sample_range, self.config.clip_sample_range)
        b_prev = 1 - a_prev
        variance = (b_prev / b_t) * (1 - a_t / a_prev)
        std = eta * variance ** (0.5)
        direction = (1 - a_prev - std**2) ** (0.5) * eps
        prev = a_prev ** (0.5) * x0 + direction
        if eta > 0:
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype)
            prev = prev + std * variance_noise
        return {"prev_sample": prev, "pred_original_sample": x0}
