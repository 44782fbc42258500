"""Analytic known-answer anchors that pin the oracle (SURVEY.md §8(c) "Substitute anchors").

The reference ships no tests/fixtures and diffusers/librosa cannot be imported here
(parity unpinned), so these derived facts are what the oracle is checked against."""
import numpy as np
import torch

from oracle.mel import Mel, mel_filterbank, griffinlim
from oracle.pipeline import AudioDiffusionPipeline
from oracle.schedulers import DDIMScheduler, DDPMScheduler
from oracle.unet import UNet2DModel, remap_deprecated_attention_keys


def test_unet_param_count():  # anchor (1): scripts/train_unet.py:115-137 config, 1 channel
    m = UNet2DModel()
    assert sum(p.numel() for p in m.parameters()) == 113_668_609
    m3 = UNet2DModel(in_channels=3, out_channels=3)
    assert sum(p.numel() for p in m3.parameters()) == 113_673_219


def test_deprecated_attention_key_remap():  # audiodiffusion/utils.py:41-54 naming
    sd = {"mid_block.attentions.0.query.weight": 1, "mid_block.attentions.0.proj_attn.bias": 2,
          "down_blocks.0.resnets.0.conv1.weight": 3}
    out = remap_deprecated_attention_keys(sd)
    assert set(out) == {"mid_block.attentions.0.to_q.weight", "mid_block.attentions.0.to_out.0.bias",
                        "down_blocks.0.resnets.0.conv1.weight"}


def test_mel_geometry_and_filterbank():  # anchor (2)
    m = Mel()
    assert m.slice_size == 131071
    fb = mel_filterbank(22050, 2048, 256)
    nz = fb > 0
    assert nz.sum() == 2032
    assert list(np.nonzero(nz[0])[0]) == [1, 2]
    assert np.nonzero(nz[255])[0][0] == 998 and np.nonzero(nz[255])[0][-1] == 1023
    assert nz.sum(1).min() == 2 and nz.sum(1).max() == 27


def test_mel_silence_is_255_and_sizes():  # audio_to_images.py:44,46-48
    m = Mel()
    m.load_audio(raw_audio=np.zeros(1000, np.float32))
    assert m.audio.dtype == np.float64 and len(m.audio) == 256 * 512  # mel.py:105-106 promotion
    im = m.audio_slice_to_image(0)
    assert im.size == (256, 256)
    assert (np.asarray(im) == 255).all()


def test_mel_sine_lights_expected_rows():
    m = Mel()
    sr, n_fft = 22050, 2048
    k = 100
    f = k * sr / n_fft
    y = np.sin(2 * np.pi * f * np.arange(m.slice_size) / sr).astype(np.float32)
    m.load_audio(raw_audio=y)
    img = np.asarray(m.audio_slice_to_image(0))
    fb = mel_filterbank(sr, n_fft, 256)
    rows = np.nonzero(fb[:, k] > 0)[0]
    assert img[:, 128].argmax() in rows


def test_image_to_audio_length_and_nnls_is_pinv_clip():
    m = Mel(x_res=64, y_res=64, hop_length=1024)
    rng = np.random.default_rng(0)
    from PIL import Image
    im = Image.fromarray(rng.integers(0, 256, (64, 64), dtype=np.uint8))
    info = []
    mag = m.image_to_stft_magnitude(im, info)
    assert all(d["nit"] == 0 and d["warnflag"] == 0 for d in info)  # L-BFGS-B stops at x0 (pgtol)
    assert np.array_equal(mag, m.image_to_stft_magnitude(im, lbfgs=False))
    a = griffinlim(mag, 4, 1024, 2048, init_phase=rng.random(mag.shape))
    assert a.shape == (1024 * 63,) and a.dtype == np.float32


def test_scheduler_tables():  # anchor (3)
    d = DDIMScheduler()
    d.set_timesteps(50)
    assert d.timesteps.tolist() == list(range(980, -1, -20))
    p = DDPMScheduler()
    p.set_timesteps(10)
    assert p.timesteps.tolist() == list(range(900, -1, -100))
    ac = d.alphas_cumprod
    np.testing.assert_allclose(ac[[0, 20, 980, 999]].numpy(), [0.99990, 0.993735, 5.9038e-5, 4.0358e-5], rtol=2e-5)


def test_ddim_eta1_full_steps_has_ddpm_variance():  # README.md:166
    d, p = DDIMScheduler(), DDPMScheduler()
    d.set_timesteps(1000), p.set_timesteps(1000)
    g = torch.Generator().manual_seed(0)
    x, e, n = (torch.randn(2, 1, 8, 8, generator=g) for _ in range(3))
    for t in (999, 500, 1):
        a = d.step(e, t, x, eta=1.0, variance_noise=n)["prev_sample"]
        b = p.step(e, t, x, variance_noise=n)["prev_sample"]
        # identical when x0 is not clipped; compare where |x0|<1
        x0 = d.step(e, t, x, eta=1.0, variance_noise=n)["pred_original_sample"]
        m = x0.abs() < 1
        assert torch.allclose(a[m], b[m], atol=2e-4)


def test_add_noise_limit_and_slerp():  # anchors (3), (5)
    d = DDIMScheduler()
    g = torch.Generator().manual_seed(0)
    x, n = torch.randn(1, 4, 4, generator=g), torch.randn(1, 4, 4, generator=g)
    # alpha_bar_0 = 0.9999: add_noise(x, n, 0) = 0.99995*x + 0.01*n
    assert float((d.add_noise(x, n, torch.tensor([0])) - x).abs().max()) <= 0.0101 * float(n.abs().max()) + 1e-4 * float(x.abs().max())
    sl = AudioDiffusionPipeline.slerp
    x0, x1 = torch.randn(16, generator=g), torch.randn(16, generator=g)
    x1 = x1 / x1.norm() * x0.norm()
    assert torch.allclose(sl(x0, x1, 0), x0) and torch.allclose(sl(x0, x1, 1), x1, atol=1e-6)
    assert abs(float(sl(x0, x1, 0.3).norm() - x0.norm())) < 1e-4


def test_rounding_rules():  # anchor (6): P5 half-to-even vs M5 trunc(x+0.5)
    assert (np.array([0.5, 127.5]).round().astype("uint8").tolist()) == [0, 128]
    assert ((np.array([0.5, 127.5]) + 0.5).astype(np.uint8).tolist()) == [1, 128]
