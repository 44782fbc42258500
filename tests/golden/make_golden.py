#!/usr/bin/env python
"""Generates the committed golden vectors (tests/golden/*.npz) from the CPU oracle.

The reference (teticio/audio-diffusion) cannot be imported in the build container (diffusers / librosa are not
installed and not vendored), so these fixtures are produced by oracle/ — the restatement pinned by the analytic
anchors of tests/test_oracle_anchors.py — NOT by the reference itself ("parity unpinned", see oracle/__init__.py).
They freeze the oracle's outputs so that (a) drift of the oracle is detected and (b) the `-m gpu` tests can compare the
HIP path against committed numbers. Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import mel as omel  # noqa: E402
from oracle import pipeline as opipe  # noqa: E402
from oracle import schedulers as osched  # noqa: E402
from oracle.unet import UNet2DModel  # noqa: E402

UNET_CFG = dict(sample_size=16, in_channels=1, out_channels=1, layers_per_block=1, block_out_channels=(32, 32),
                down_block_types=("DownBlock2D", "AttnDownBlock2D"), up_block_types=("AttnUpBlock2D", "UpBlock2D"))
MEL_CFG = dict(x_res=32, y_res=32, hop_length=128, n_fft=512, n_iter=4, sample_rate=8000)


def main():
    torch.manual_seed(1234)
    unet = UNet2DModel(**UNET_CFG).eval()
    with torch.no_grad():
        for n, p in unet.named_parameters():
            if "norm" in n:
                p.add_(0.1 * torch.randn_like(p))
    g = torch.Generator().manual_seed(99)
    x = torch.randn(2, 1, 16, 16, generator=g)
    out = {"x": x.numpy()}
    with torch.no_grad():
        for t in (980, 37):
            out[f"eps_t{t}"] = unet(x, torch.tensor(t))["sample"].numpy()
    pipe = opipe.AudioDiffusionPipeline(None, unet, omel.Mel(**MEL_CFG), osched.DDIMScheduler())
    imgs, final = pipe(batch_size=2, steps=4, noise=x.clone(), audio=False, return_float=True)
    out["ddim4_final"] = final.numpy()
    out["ddim4_u8"] = np.stack([np.asarray(i) for i in imgs])
    sn = [torch.randn(2, 1, 16, 16, generator=g) for _ in range(3)]
    pipe2 = opipe.AudioDiffusionPipeline(None, unet, omel.Mel(**MEL_CFG), osched.DDPMScheduler())
    _, final2 = pipe2(batch_size=2, steps=3, noise=x.clone(), step_noise=sn, audio=False, return_float=True)
    out["ddpm3_final"] = final2.numpy()
    out["ddpm3_step_noise"] = np.stack([s.numpy() for s in sn])
    for k, v in unet.state_dict().items():
        out["w:" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "unet_tiny.npz"), **out)

    # scheduler tables (diffusers formulas restated in oracle/schedulers.py): alphas_cumprod + timesteps
    d, p = osched.DDIMScheduler(), osched.DDPMScheduler()
    d.set_timesteps(50), p.set_timesteps(1000)
    np.savez_compressed(os.path.join(HERE, "sched_tables.npz"), alphas_cumprod=d.alphas_cumprod.numpy(),
                        ddim50_timesteps=d.timesteps.numpy(), ddpm1000_timesteps=p.timesteps.numpy())

    # mel codec
    rng = np.random.default_rng(7)
    m = omel.Mel(**MEL_CFG)
    tt = np.arange(m.slice_size) / MEL_CFG["sample_rate"]
    y = (0.1 * rng.standard_normal(m.slice_size) + 0.5 * np.sin(2 * np.pi * 440 * tt) + 0.3 * np.sin(2 * np.pi * 1500 * tt)).astype(np.float32)
    m.load_audio(raw_audio=y)
    img = m.audio_slice_to_image(0)
    phase = rng.random((1 + MEL_CFG["n_fft"] // 2, MEL_CFG["x_res"]))
    mag = m.image_to_stft_magnitude(img)
    audio = omel.griffinlim(mag, m.n_iter, m.hop_length, m.n_fft, init_phase=phase)
    fb = omel.mel_filterbank(MEL_CFG["sample_rate"], MEL_CFG["n_fft"], MEL_CFG["y_res"])
    starts = np.array([np.nonzero(r > 0)[0][0] for r in fb], np.int32)
    counts = np.array([np.nonzero(r > 0)[0][-1] - np.nonzero(r > 0)[0][0] + 1 for r in fb], np.int32)
    np.savez_compressed(os.path.join(HERE, "mel_small.npz"), audio_in=y, image=np.asarray(img), init_phase=phase,
                        stft_mag=mag, audio_out=audio, fb_start=starts, fb_count=counts)
    print("wrote", [f for f in os.listdir(HERE) if f.endswith(".npz")])


if __name__ == "__main__":
    main()
