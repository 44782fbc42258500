"""Timings of BASELINE.json's other configurations on one MI355X (not the bench metric): config 2 = pixel-space DDPM-1000,
B=16, 256x256; config 4 = latent DDPM-1000 (UNet at 32x32 on VAE latents) + VAE decode, B=16; config 1 = 64x64 DDPM-10, B=1."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
from audiodiffusion import AudioDiffusionPipeline, DDPMScheduler, Mel, UNet2DModel, _native  # noqa: E402
from audiodiffusion.vae import AutoencoderKL  # noqa: E402

_native.load()
dev = torch.device("cuda:0")


def unet_cfg(res, ch=1):
    return dict(sample_size=(res, res), in_channels=ch, out_channels=ch, layers_per_block=2,
                block_out_channels=(128, 128, 256, 256, 512, 512),
                down_block_types=("DownBlock2D",) * 4 + ("AttnDownBlock2D", "DownBlock2D"),
                up_block_types=("UpBlock2D", "AttnUpBlock2D") + ("UpBlock2D",) * 4)


def run(name, pipe, B, steps, res, reps=1):
    g = torch.Generator(device="cpu").manual_seed(42)
    noise = torch.randn(B, pipe.unet.config["in_channels"], res, res, generator=g).to(dev)
    pipe(batch_size=B, steps=min(steps, 4), noise=noise.clone(), audio=False)          # warm-up / graph capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        pipe(batch_size=B, steps=steps, noise=noise.clone(), audio=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name}: B={B} steps={steps}: {dt:.2f} s per batch = {B / dt:.3f} spectrograms/s ({dt / steps * 1e3:.2f} ms/step)", flush=True)


which = sys.argv[1:] or ["c1", "c2", "c4"]
if "c1" in which:
    p = AudioDiffusionPipeline(None, UNet2DModel(**unet_cfg(64)).init_random(0), Mel(x_res=64, y_res=64, hop_length=1024), DDPMScheduler())
    p.set_progress_bar_config(disable=True)
    run("config 1 (64x64, DDPM-10)", p, 1, 10, 64, reps=3)
if "c2" in which:
    p = AudioDiffusionPipeline(None, UNet2DModel(**unet_cfg(256)).init_random(0), Mel(), DDPMScheduler())
    p.set_progress_bar_config(disable=True)
    run("config 2 (256x256 pixel-space, DDPM-1000)", p, 16, 1000, 256)
if "c4" in which:
    vae = AutoencoderKL(sample_size=(256, 256), in_channels=1, out_channels=1, latent_channels=1, layers_per_block=2,
                        block_out_channels=(128, 256, 512, 512), down_block_types=("DownEncoderBlock2D",) * 4,
                        up_block_types=("UpDecoderBlock2D",) * 4).init_random(0)
    p = AudioDiffusionPipeline(vae, UNet2DModel(**unet_cfg(32)).init_random(1), Mel(), DDPMScheduler())
    p.set_progress_bar_config(disable=True)
    run("config 4 (latent 32x32 + VAE decode to 256x256, DDPM-1000)", p, 16, 1000, 32)
