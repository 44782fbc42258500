"""DDPMScheduler / DDIMScheduler drop-ins (duck-typed members the reference pipeline calls).

Reference call sites: `audiodiffusion/pipeline_audio_diffusion.py:115,150,157,166-179,221-234`,
`scripts/train_unet.py:161-164,250`. Host side (this file): the beta/alpha tables and the per-step scalar
coefficients, computed in the same 0-d fp32 tensor arithmetic diffusers==0.24.0 uses. Device side: ONE fused
HIP kernel per step (`adm_sched_step`, csrc/k_sched.hip) instead of ~12 eager elementwise kernels plus D2H
scalar reads. `coef_rows()` exports the whole coefficient table so the native sampling loop
(`adm_sample_loop`) can replay a captured hipGraph for every step.
"""
import json
import math
import os

import numpy as np
import torch

from . import ops


class FrozenConfig(dict):
    """dict with attribute access, like diffusers' FrozenDict."""
    __getattr__ = dict.__getitem__


class SchedulerOutput(dict):
    __getattr__ = dict.__getitem__


def randn_tensor(shape, generator=None, device=None, dtype=torch.float32):
    """diffusers.utils.torch_utils.randn_tensor: a CPU generator draws on the CPU and the result is moved."""
    device = torch.device(device or "cpu")
    gen_dev = generator.device.type if generator is not None else device.type
    if gen_dev == "cpu" and device.type != "cpu":
        return torch.randn(shape, generator=generator, device="cpu", dtype=dtype).to(device)
    return torch.randn(shape, generator=generator, device=device, dtype=dtype)


def _betas(n, beta_start, beta_end, schedule):
    if schedule == "linear":
        return torch.linspace(beta_start, beta_end, n, dtype=torch.float32)
    if schedule == "scaled_linear":
        return torch.linspace(beta_start**0.5, beta_end**0.5, n, dtype=torch.float32) ** 2
    if schedule == "squaredcos_cap_v2":
        f = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2  # noqa: E731
        return torch.tensor([min(1 - f((i + 1) / n) / f(i / n), 0.999) for i in range(n)], dtype=torch.float32)
    raise NotImplementedError(f"{schedule} is not implemented")


class _SchedulerBase:
    config_name = "scheduler_config.json"
    _class_name = "SchedulerMixin"
    _defaults = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                     trained_betas=None, clip_sample=True, clip_sample_range=1.0, prediction_type="epsilon",
                     timestep_spacing="leading", steps_offset=0, thresholding=False)

    def __init__(self, **kwargs):
        cfg = dict(self._defaults)
        unknown = {k: v for k, v in kwargs.items() if k not in cfg and not k.startswith("_")}
        cfg.update({k: v for k, v in kwargs.items() if not k.startswith("_")})
        self._unknown = unknown
        if cfg["prediction_type"] != "epsilon":
            raise NotImplementedError("only prediction_type='epsilon' (what the reference trains) is implemented")
        if cfg["timestep_spacing"] != "leading":
            raise NotImplementedError("only timestep_spacing='leading' is implemented")
        if cfg.get("thresholding"):
            raise NotImplementedError("thresholding is not implemented")
        self.config = FrozenConfig(cfg)
        if cfg["trained_betas"] is not None:
            self.betas = torch.tensor(cfg["trained_betas"], dtype=torch.float32)
        else:
            self.betas = _betas(cfg["num_train_timesteps"], cfg["beta_start"], cfg["beta_end"], cfg["beta_schedule"])
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, cfg["num_train_timesteps"])[::-1].copy())
        self._table = None  # (device, eta) -> device coefficient table

    # ---- config (de)serialisation: scheduler/scheduler_config.json of the diffusers layout -------------
    @classmethod
    def from_config(cls, cfg):
        return cls(**{k: v for k, v in dict(cfg).items() if not k.startswith("_")})

    @classmethod
    def from_pretrained(cls, path, subfolder=None):
        p = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(p, cls.config_name)) as f:
            return cls.from_config(json.load(f))

    def save_pretrained(self, path):
        os.makedirs(path, exist_ok=True)
        d = {"_class_name": self._class_name, "_diffusers_version": "0.24.0"}
        d.update(self.config)
        with open(os.path.join(path, self.config_name), "w") as f:
            json.dump(d, f, indent=2, sort_keys=True)

    # ---- reference members -----------------------------------------------------------------------------
    def set_timesteps(self, num_inference_steps, device=None):
        n_train = self.config.num_train_timesteps
        if num_inference_steps > n_train:
            raise ValueError("num_inference_steps cannot exceed num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        step_ratio = n_train // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
        ts += self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)
        self._table = None

    def scale_model_input(self, sample, timestep=None):
        return sample

    def add_noise(self, original_samples, noise, timesteps):
        """sqrt(acp[t])*x0 + sqrt(1-acp[t])*noise with diffusers' broadcasting; the two shapes the reference
        uses (pipeline:150,157; train_unet.py:250) run in the fused HIP kernel."""
        ac = self.alphas_cumprod
        ts = torch.as_tensor(timesteps).cpu().long()
        sa = (ac[ts] ** 0.5).flatten().to(noise.device).contiguous()
        sb = ((1 - ac[ts]) ** 0.5).flatten().to(noise.device).contiguous()
        x0 = original_samples.contiguous()
        nz = noise.contiguous()
        if nz.dim() == 4 and x0.shape == nz.shape and sa.numel() == nz.shape[0]:
            return ops.add_noise(x0, nz, sa, sb, per_sample=True)
        if nz.dim() == 4 and x0.dim() == 3 and nz.shape[1] == 1 and x0.shape[0] == 1 and ts.dim() == 1:
            return ops.add_noise(x0, nz, sa, sb, per_sample=False)  # (B, n_steps, H, W) mask (pipeline:157)
        if ts.dim() == 0 and x0.shape[-2:] == nz.shape[-2:] and nz.dim() == 4:
            # pipeline:150: (1,H,W) x (B,1,H,W) with one timestep -> (B,1,H,W)
            return ops.add_noise(x0.reshape(1, *x0.shape[-2:]).contiguous(), nz, sa, sb, per_sample=False)
        raise NotImplementedError("add_noise: unsupported broadcast pattern for the fused kernel")

    def _index_of(self, timestep):
        t = int(timestep)
        idx = (self.timesteps == t).nonzero()
        if idx.numel() == 0:
            raise ValueError(f"timestep {t} is not in the current schedule")
        return int(idx[0])

    def coef_table(self, device, eta=0.0):
        return self._cached(device, eta)[1]

    def _cached(self, device, eta):
        key = (str(device), float(eta))
        if self._table is None or self._table[0] != key:
            rows = self.coef_rows(eta)
            self._table = (key, ops.sched_coef_table(rows, device), rows)
        return self._table

    def _prev_acp(self, t):
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        return self.alphas_cumprod[prev_t] if prev_t >= 0 else self._final_alpha()

    def _clip(self):
        return float(self.config.clip_sample_range) if self.config.clip_sample else -1.0

    def _step(self, model_output, timestep, sample, eta, generator, variance_noise):
        i = self._index_of(timestep)
        _, table, rows = self._cached(sample.device, eta)
        need_noise = rows[i]["k_noise"] != 0.0
        if need_noise and variance_noise is None:
            variance_noise = randn_tensor(model_output.shape, generator, model_output.device, model_output.dtype)
        prev = ops.sched_step(sample.contiguous(), model_output.contiguous(), table, i,
                              noise=variance_noise.contiguous() if variance_noise is not None else None)
        return SchedulerOutput(prev_sample=prev)


class DDPMScheduler(_SchedulerBase):
    _class_name = "DDPMScheduler"
    _defaults = dict(_SchedulerBase._defaults, variance_type="fixed_small", clip_sample=True)

    def __init__(self, **kw):
        super().__init__(**kw)
        if self.config.variance_type != "fixed_small":
            raise NotImplementedError("only variance_type='fixed_small' is implemented")

    def _final_alpha(self):
        return self.one

    def coef_rows(self, eta=0.0):
        rows = []
        for t in self.timesteps.tolist():
            a_t = self.alphas_cumprod[t]
            a_prev = self._prev_acp(t)
            b_t = 1 - a_t
            b_prev = 1 - a_prev
            cur_a = a_t / a_prev
            cur_b = 1 - cur_a
            k_noise = 0.0
            if t > 0:
                var = torch.clamp((1 - a_prev) / (1 - a_t) * cur_b, min=1e-20)
                k_noise = float(var ** 0.5)
            rows.append(dict(sqrt_beta=float(b_t ** 0.5), sqrt_alpha=float(a_t ** 0.5), clip=self._clip(),
                             k_x0=float((a_prev ** 0.5 * cur_b) / b_t), k_x=float(cur_a ** 0.5 * b_prev / b_t),
                             k_eps=0.0, k_noise=k_noise, timestep=float(t)))
        return rows

    def step(self, model_output, timestep, sample, generator=None, return_dict=True, variance_noise=None):
        return self._step(model_output, timestep, sample, 0.0, generator, variance_noise)


class DDIMScheduler(_SchedulerBase):
    _class_name = "DDIMScheduler"
    _defaults = dict(_SchedulerBase._defaults, set_alpha_to_one=True, clip_sample=True)

    def __init__(self, **kw):
        super().__init__(**kw)
        self.final_alpha_cumprod = torch.tensor(1.0) if self.config.set_alpha_to_one else self.alphas_cumprod[0]

    def _final_alpha(self):
        return self.final_alpha_cumprod

    def coef_rows(self, eta=0.0):
        rows = []
        for t in self.timesteps.tolist():
            a_t = self.alphas_cumprod[t]
            a_prev = self._prev_acp(t)
            b_t = 1 - a_t
            b_prev = 1 - a_prev
            variance = (b_prev / b_t) * (1 - a_t / a_prev)
            std = eta * variance ** 0.5
            rows.append(dict(sqrt_beta=float(b_t ** 0.5), sqrt_alpha=float(a_t ** 0.5), clip=self._clip(),
                             k_x0=float(a_prev ** 0.5), k_x=0.0, k_eps=float((1 - a_prev - std ** 2) ** 0.5),
                             k_noise=float(std) if eta > 0 else 0.0, timestep=float(t)))
        return rows

    def encode_rows(self):
        """Coefficients of the DDIM inversion update, `pipeline_audio_diffusion.py:228-240`, in loop order
        (ascending timesteps): x = (x - c_dir*eps) * a_prev^-0.5 * a_t^0.5 + b_t^0.5 * eps."""
        rows = []
        for t in torch.flip(self.timesteps, (0,)).tolist():
            a_t = self.alphas_cumprod[t]
            a_prev = self._prev_acp(t)
            b_t = 1 - a_t
            rows.append(dict(sqrt_beta=float((1 - a_prev) ** 0.5), sqrt_alpha=float(a_prev ** (-0.5)), clip=-1.0,
                             k_x0=float(a_t ** 0.5), k_x=0.0, k_eps=float(b_t ** 0.5), k_noise=0.0, timestep=float(t)))
        return rows

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        if use_clipped_model_output:
            raise NotImplementedError("use_clipped_model_output is not implemented (the reference never sets it)")
        return self._step(model_output, timestep, sample, float(eta), generator, variance_noise)
