"""Long-form / interpolation procedures of notebooks/test_model.ipynb: `audiodiffusion.longform` on the
native pipeline vs the notebook cells run, cell by cell, on the oracle pipeline — same weights, same noise, same injected
per-step noise and Griffin-Lim phases. Bars: images <= 1 LSB and >= 97 % identical (16x16 toy images: one pixel is 0.4 %, and
a 1e-6 difference before the rounding flips one now and then), audio <= 2e-3 of its peak when the images are identical and
<= 1e-2 for the chained tracks (the audio of segment k conditions segment k+1; one flipped image LSB = 0.31 dB in one mel bin moves
a segment's Griffin-Lim output by up to ~0.5 % of its peak, measured on the MI355X), slerp <= 1e-6."""
import numpy as np
import pytest
import torch

from native_backend import BACKENDS, select
from oracle import mel as omel
from oracle import pipeline as opipe
from oracle import schedulers as osched
from oracle.unet import UNet2DModel as OracleUNet

TINY = dict(sample_size=(16, 16), in_channels=1, out_channels=1, layers_per_block=1, block_out_channels=(32, 64),
            down_block_types=("DownBlock2D", "AttnDownBlock2D"), up_block_types=("AttnUpBlock2D", "UpBlock2D"))
MEL = dict(x_res=16, y_res=16, hop_length=64, n_fft=256, n_iter=2)
SR = 22050
N_BINS = 1 + MEL["n_fft"] // 2


def _pipes(dev, kind="ddim"):
    from audiodiffusion import AudioDiffusionPipeline, DDIMScheduler, DDPMScheduler, Mel, UNet2DModel
    torch.manual_seed(0)
    ref_unet = OracleUNet(**TINY).eval()
    unet = UNet2DModel(**TINY).load_state_dict(ref_unet.state_dict())
    mine = AudioDiffusionPipeline(None, unet, Mel(**MEL), (DDIMScheduler if kind == "ddim" else DDPMScheduler)()).to(dev)
    mine.set_progress_bar_config(disable=True)
    ref = opipe.AudioDiffusionPipeline(None, ref_unet, omel.Mel(**MEL),
                                       (osched.DDIMScheduler if kind == "ddim" else osched.DDPMScheduler)())
    return mine, ref


def _clip(n, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / SR
    return (0.2 * rng.standard_normal(n) + 0.4 * np.sin(2 * np.pi * 1500.0 * t)).astype(np.float32)


def _same_images(a, b, frac=0.97):
    a, b = np.asarray(a).astype(int), np.asarray(b).astype(int)
    assert a.shape == b.shape and np.abs(a - b).max() <= 1 and (a == b).mean() >= frac, (np.abs(a - b).max(), (a == b).mean())


@pytest.mark.parametrize("backend", BACKENDS)
def test_slerp_grid_matches_the_reference_expression(backend):
    dev = select(backend)
    from audiodiffusion import AudioDiffusionPipeline
    from audiodiffusion.longform import slerp_grid
    g = torch.Generator().manual_seed(1)
    x0, x1 = torch.randn(1, 1, 16, 16, generator=g), torch.randn(1, 1, 16, 16, generator=g)
    alphas = [0.0, 0.1, 0.5, 0.9, 1.0]
    grid = slerp_grid(x0.to(dev), x1.to(dev), alphas).cpu()
    assert grid.shape == (5, 1, 1, 16, 16)
    for a, got in zip(alphas, grid):
        want = opipe.AudioDiffusionPipeline.slerp(x0, x1, a)           # pipeline_audio_diffusion.py:244-258 on torch-CPU
        assert float((got - want).abs().max()) <= 1e-6
        assert float((got.to(dev) - AudioDiffusionPipeline.slerp(x0.to(dev), x1.to(dev), a)).abs().max()) <= 1e-6
    assert float((grid[0] - x0).abs().max()) <= 1e-6 and float((grid[-1] - x1).abs().max()) <= 1e-6


@pytest.mark.parametrize("backend", BACKENDS)
def test_interpolate_encode_slerp_sample(backend):
    """cells 32-37: encode two images, slerp, sample — batched here, one call per image / alpha on the oracle."""
    dev = select(backend)
    from audiodiffusion.longform import interpolate
    mine, ref = _pipes(dev)
    g = torch.Generator().manual_seed(3)
    imgs = ref(batch_size=2, steps=3, noise=torch.randn(2, 1, 16, 16, generator=g), audio=False, return_dict=False)[0]
    alphas = [0.25, 0.5, 0.75]
    phases = np.random.default_rng(0).random((3, N_BINS, 16))
    images, (sr, audios) = interpolate(mine, imgs[0], imgs[1], alphas, steps=4, encode_steps=5, init_phases=phases)
    n0, n1 = ref.encode([imgs[0]], steps=5), ref.encode([imgs[1]], steps=5)
    assert sr == SR and len(images) == len(audios) == 3
    for i, a in enumerate(alphas):
        ri, (_, ra) = ref(batch_size=1, steps=4, noise=ref.slerp(n0, n1, a), return_dict=False, init_phase=phases[i:i + 1])
        _same_images(images[i], ri[0])
        if (np.asarray(images[i]) == np.asarray(ri[0])).all():
            got = audios[i]
        else:       # an LSB of the image flipped (a 1e-7 difference before the rounding): the codec is compared on the oracle's image
            got = mine.mel.image_to_audio(ri[0], init_phase=phases[i])
        assert np.abs(got - ra[0]).max() <= 2e-3 * max(np.abs(ra[0]).max(), 1e-6)


def _ref_one(ref, **kw):
    images, (sr, audios) = ref(batch_size=1, return_dict=False, **kw)
    return images[0], sr, audios[0]


@pytest.mark.parametrize("backend", BACKENDS)
def test_outpaint_chain_matches_the_notebook_cell_on_the_oracle(backend):
    """cell 16 with DDPM steps (per-step noise injected): three continuations, each pinned to the previous tail."""
    dev = select(backend)
    from audiodiffusion.longform import outpaint
    mine, ref = _pipes(dev, "ddpm")
    n_seg, steps, ov_secs = 3, 4, 256 / SR
    g = torch.Generator().manual_seed(5)
    noise = [torch.randn(1, 1, 16, 16, generator=g) for _ in range(n_seg)]
    step_noise = [torch.randn(steps, 1, 1, 16, 16, generator=g) for _ in range(n_seg)]
    phases = np.random.default_rng(2).random((n_seg, 1, N_BINS, 16))
    start = _clip(1024, 7)
    track, images = outpaint(mine, start, n_seg, ov_secs, steps=steps, noise=[x.to(dev) for x in noise],
                             step_noise=[x.to(dev) for x in step_noise], init_phases=phases)
    # the notebook cell, on the oracle
    ov = int(ov_secs * SR)
    rtrack, audio = start, start
    for i in range(n_seg):
        rimg, _, audio2 = _ref_one(ref, raw_audio=audio[-ov:], start_step=0, steps=steps, mask_start_secs=ov_secs,
                                   noise=noise[i].clone(), step_noise=step_noise[i], init_phase=phases[i])
        _same_images(images[i], rimg)
        rtrack = np.concatenate([rtrack, audio2[ov:]])
        audio = audio2
    assert track.shape == rtrack.shape == (1024 + n_seg * (960 - ov),)
    assert np.abs(track - rtrack).max() <= 1e-2 * np.abs(rtrack).max()
    # the pinned columns of every generated image really are the (noised-to-step-0) input: first 4 px of 16
    assert int(ov_secs * SR / MEL["hop_length"]) == 4


@pytest.mark.parametrize("backend", BACKENDS)
def test_remix_track_matches_the_notebook_cell_on_the_oracle(backend):
    """cell 20: overlapping slices from start_step, generated tail re-inserted (peak-normalised) into the next slice."""
    dev = select(backend)
    from audiodiffusion.longform import remix_track
    mine, ref = _pipes(dev, "ddim")
    steps, start_step, ov_secs = 6, 3, 256 / SR
    noise = torch.randn(1, 1, 16, 16, generator=torch.Generator().manual_seed(9))
    audio_in = _clip(1024 * 3, 11)
    n_slices = len(audio_in) // (1024 - 256)
    phases = np.random.default_rng(4).random((n_slices, 1, N_BINS, 16))
    track, images = remix_track(mine, audio_in, ov_secs, start_step, seed=1, steps=steps, noise=noise.to(dev), init_phases=phases)
    ov, stride = 256, 1024 - 256
    rtrack, audio2, not_first = np.array([]), None, 0
    for s in range(n_slices):
        audio = np.array(audio_in[s * stride:s * stride + 1024])
        if not_first:
            audio[:ov] = audio2[-ov:] * np.max(audio[:ov]) / np.max(audio2[-ov:])
        rimg, _, audio2 = _ref_one(ref, raw_audio=audio, start_step=start_step, steps=steps, mask_start_secs=ov_secs * not_first,
                                   noise=noise.clone(), init_phase=phases[s])
        _same_images(images[s], rimg)
        rtrack = np.concatenate([rtrack, audio2[ov * not_first:]])
        not_first = 1
    assert len(images) == n_slices == 4 and track.shape == rtrack.shape
    assert np.abs(track - rtrack).max() <= 1e-2 * np.abs(rtrack).max()
