"""Test helper: a BRIEFLY TRAINED denoiser for end-to-end sampler parity.

With random weights the DDIM sampler is chaotic (a 1e-6 perturbation of the start noise grows to O(1) within 20-30 steps, on
the oracle itself), so two correct fp32 implementations of the 50-step loop end in different images. A denoiser that has seen
even ~100 optimizer steps of a structured data set is contractive (the same perturbation stays at ~5e-6 over all 50 steps:
measured in tools/ddim50_parity.py and profiles/r03_ddim50_parity.md), which is what makes a complete DDIM-50 comparison
against the oracle meaningful — `pipeline_audio_diffusion.py:69,159-185` end to end.

synthetic_mel : mel-spectrogram-like images in [-1, 1] (a harmonic stack with vibrato and an amplitude envelope per sample).
train_oracle  : a few AdamW steps of the scripts/train_unet.py objective on the torch-CPU oracle (tiny models, emulator tests).
train_product : the same objective through the product's native trainer (full-size model on the MI355X).
"""
import math

import torch


def synthetic_mel(B, hw, g):
    H, W = hw
    yy = torch.linspace(0, 1, H).view(1, H, 1)
    xx = torch.linspace(0, 1, W).view(1, 1, W)
    img = torch.full((B, H, W), -1.0)
    f0 = torch.rand(B, 1, 1, generator=g) * 0.15 + 0.05
    width = max(0.01, 1.0 / H)
    for h in range(1, 6):
        ph, fr = torch.rand(B, 1, 1, generator=g), torch.rand(B, 1, 1, generator=g)
        env = torch.rand(B, 1, 1, generator=g)
        center = f0 * h + 0.02 * torch.sin(2 * math.pi * (xx * fr * 3 + ph))
        amp = (0.5 + 0.5 * torch.sin(2 * math.pi * (xx * env * 4 + ph))) / h
        img = img + 2 * amp * torch.exp(-((yy - center) / width) ** 2)
    return img.clamp(-1, 1).unsqueeze(1).contiguous()


def train_oracle(model, hw, steps, batch=8, lr=1e-3, seed=2):
    """scripts/train_unet.py:250-262 on the oracle with torch autograd (AdamW, clip 1.0); returns the losses."""
    from oracle import schedulers as osched
    ns = osched.DDPMScheduler()
    opt = torch.optim.AdamW(model.parameters(), lr=lr, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)
    g = torch.Generator().manual_seed(seed)
    losses = []
    model.train()
    for _ in range(steps):
        clean = synthetic_mel(batch, hw, g)
        noise = torch.randn(clean.shape, generator=g)
        ts = torch.randint(0, 1000, (batch,), generator=g)
        loss = torch.nn.functional.mse_loss(model(ns.add_noise(clean, noise, ts), ts)["sample"], noise)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        losses.append(float(loss.detach()))
    model.eval()
    return losses


def train_product(unet, hw, steps, dev, batch=16, lr=1e-4, mixed_precision="bf16", seed=2, on_step=None):
    """The product's training step (native forward + backward, clip, fused AdamW, weight re-pack) on synthetic_mel batches;
    the trained weights end up in unet.state_dict(). Returns the losses."""
    from audiodiffusion import DDPMScheduler
    from audiodiffusion import training as T
    flat, grads = unet.enable_training(hw, mixed_precision=mixed_precision)
    opt = T.AdamW(flat, lr=lr)
    ns = DDPMScheduler()
    g = torch.Generator().manual_seed(seed)
    losses = []
    for i in range(steps):
        clean = synthetic_mel(batch, hw, g).to(dev)
        noise = torch.randn(clean.shape, generator=g).to(dev)
        ts = torch.randint(0, 1000, (batch,), generator=g)
        loss = unet.train_step(ns.add_noise(clean, noise, ts), ts, noise)
        opt.step(grads, clip=T.clip_grad_norm_(grads, 1.0))
        unet.refresh_weights()
        losses.append(loss)
        if on_step is not None:
            on_step(i + 1)
    unet.sync_state_dict_from_flat()
    return [float(v) for v in losses]
