"""Checks whose two sides do NOT share source text.

The product (`audiodiffusion/mel.py`, `schedulers.py`) and the oracle (`oracle/mel.py`, `oracle/schedulers.py`) both restate
librosa / diffusers, so comparing one with the other proves little where the two files read alike.  Every expected value
in this file is derived a third way, inside the test: the Slaney filter corner frequencies from the scalar closed form of
the auditory scale, spectra from `torch.stft` / `torch.istft`, the diffusion tables from float64 closed forms of
Ho et al. (DDPM) / Song et al. (DDIM).  The mel-bin indexing check goes THROUGH the HIP kernel (C-ABI
`adm_mel_forward_power`): north_star's one bit-exact requirement.
"""
import math

import numpy as np
import pytest
import torch

from native_backend import BACKENDS, select
from oracle import mel as omel
from oracle import schedulers as osched

SR, N_FFT, N_MELS, HOP = 22050, 2048, 256, 512


# ------------------------------------------------------------------------------------------------ Slaney corners
def slaney_corner_hz(sr, n_mels):
    """Corner frequencies of the n_mels triangular filters, scalar math only: n_mels + 2 points equally spaced on
    Slaney's scale between 0 Hz and sr/2, where the scale is f / (200/3) below 1 kHz and 15 + 27 * log_6.4(f / 1000)
    above (Slaney, Auditory Toolbox 1998)."""
    top_hz = sr / 2.0
    top = top_hz * 3.0 / 200.0 if top_hz < 1000.0 else 15.0 + 27.0 * math.log(top_hz / 1000.0, 6.4)
    out = []
    for i in range(n_mels + 2):
        z = top * i / (n_mels + 1)
        out.append(z * 200.0 / 3.0 if z < 15.0 else 1000.0 * 6.4 ** ((z - 15.0) / 27.0))
    out[0], out[-1] = 0.0, top_hz                      # the end points ARE fmin = 0 and fmax = sr/2, not round trips
    return out


def expected_support(sr, n_fft, n_mels):
    """[first_bin, last_bin] of every filter: the FFT bins strictly inside (corner[m], corner[m+2]); also returns the
    smallest relative distance of any bin to a corner (the derivation is only meaningful if no bin sits ON a corner)."""
    corner = slaney_corner_hz(sr, n_mels)
    n_bins = n_fft // 2 + 1
    hz = [k * sr / n_fft for k in range(n_bins)]
    sup, margin = [], 1.0
    for m in range(n_mels):
        lo, hi = corner[m], corner[m + 2]
        inside = [k for k in range(n_bins) if lo < hz[k] < hi]
        sup.append((inside[0], inside[-1]) if inside else None)
        for k in range(n_bins):
            for c in (lo, hi):
                if not ((k == 0 and c == 0.0) or (k == n_bins - 1 and c == sr / 2.0)):   # DC / Nyquist sit ON the end corners
                    margin = min(margin, abs(hz[k] - c) / max(c, 1.0))
    return sup, margin


def test_expected_supports_are_unambiguous_and_match_the_survey_anchors():
    sup, margin = expected_support(SR, N_FFT, N_MELS)
    assert margin > 1e-9, margin                      # no bin within rounding distance of a corner (except 0 Hz itself)
    assert sup[0] == (1, 2) and sup[255] == (998, 1023)
    assert sum(b - a + 1 for a, b in sup) == 2032


@pytest.mark.parametrize("cfg", [(22050, 2048, 256), (22050, 1024, 64), (44100, 2048, 128), (16000, 512, 40), (48000, 4096, 80)])
def test_sparse_tap_generator_vs_dense_librosa_form_vs_closed_form_corners(cfg):
    """Three derivations of the same filterbank: the product's sparse per-filter generator, the oracle's dense
    librosa-shaped matrix, and the corner closed form above (support only)."""
    from audiodiffusion.mel import slaney_filter_taps, taps_to_dense
    sr, n_fft, n_mels = cfg
    start, count, w32, w64 = slaney_filter_taps(sr, n_fft, n_mels)
    n_bins = n_fft // 2 + 1
    assert np.array_equal(taps_to_dense(start, count, w32, n_bins), omel.mel_filterbank(sr, n_fft, n_mels, np.float32))
    assert np.array_equal(taps_to_dense(start, count, w64, n_bins), omel.mel_filterbank(sr, n_fft, n_mels, np.float64))
    sup, margin = expected_support(sr, n_fft, n_mels)
    if margin > 1e-9:
        for m in range(n_mels):
            got = (int(start[m]), int(start[m] + count[m] - 1)) if count[m] else None
            if got != sup[m] and m == n_mels - 1 and got == (sup[m][0], n_bins - 1):
                # librosa's top corner is the round trip mel_to_hz(hz_to_mel(sr/2)), which can land one ulp ABOVE sr/2
                # and then gives the Nyquist bin a weight of rounding size; the product reproduces that bit for bit
                dense = taps_to_dense(start, count, w64, n_bins)
                assert 0 < dense[m, n_bins - 1] < 1e-12 * dense[m].max()
                continue
            assert got == sup[m], (m, got, sup[m])


@pytest.mark.parametrize("backend", BACKENDS)
def test_mel_bin_indexing_through_the_kernel_is_bit_exact(backend):
    """A cosine on FFT bin k under the periodic Hann window has energy in bins k-1, k, k+1 only (elsewhere rounding noise). Its mel power spectrogram, computed by the HIP kernel, must be non-zero in EXACTLY the
    filters whose closed-form support touches {k-1, k, k+1} — for every bin (hardware) / a spread of bins (emulator)."""
    select(backend)
    from audiodiffusion.mel import Mel
    m = Mel(x_res=8, y_res=N_MELS, sample_rate=SR, n_fft=N_FFT, hop_length=HOP)       # 8 frames; frames 2..5 see no padding
    sup, _ = expected_support(SR, N_FFT, N_MELS)
    bins = list(range(1, N_FFT // 2)) if backend == "hip" else [1, 2, 3, 17, 100, 257, 511, 640, 777, 901, 997, 998, 1022, 1023]
    n = m.slice_size
    t = np.arange(n, dtype=np.float64)
    for dtype in (np.float32, np.float64) if backend == "hip" else (np.float32,):
        audio = np.stack([np.cos(2 * np.pi * k * t / N_FFT) for k in bins]).astype(dtype)
        spec = m.audio_slices_to_melspectrograms(audio).astype(np.float64)            # (len(bins), 256, 8)
        assert spec.shape == (len(bins), N_MELS, 8)
        for row, k in enumerate(bins):
            want = np.array([s is not None and s[0] <= k + 1 and s[1] >= k - 1 for s in sup])
            for frame in (2, 3, 4, 5):
                col = spec[row, :, frame]
                peak = col.max()
                assert peak > 0
                # measured: genuine rows >= 2e-4 of the peak; rounding leakage <= 6e-16 (float32 audio: the cosine itself is
                # rounded to 24 bits) / 2e-24 (float64). The threshold sits between with >= 100x clearance on both sides.
                live = col > 1e-11 * peak
                dead_zone = (col > 1e-13 * peak) & (col < 1e-9 * peak)
                assert not dead_zone.any(), (k, frame, col[dead_zone])
                assert np.array_equal(live, want), (k, frame, np.nonzero(live != want)[0])


# ------------------------------------------------------------------------------------------------ STFT / iSTFT
def _noise_audio(n, dtype, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / SR
    return (0.2 * rng.standard_normal(n) + 0.4 * np.sin(2 * np.pi * 997.0 * t)).astype(dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_oracle_stft_and_istft_agree_with_torch(dtype):
    """oracle/mel.py's stft / istft (restated from librosa) against torch's independent implementation."""
    y = _noise_audio(8 * HOP - 1, dtype)
    D = omel.stft(y, N_FFT, HOP)
    win = torch.hann_window(N_FFT, periodic=True, dtype=torch.float64)
    T = torch.stft(torch.from_numpy(y.astype(np.float64)), N_FFT, HOP, window=win, center=True, pad_mode="constant",
                   return_complex=True).numpy()
    assert D.shape == T.shape == (N_FFT // 2 + 1, 8)
    scale = np.abs(T).max()
    assert np.abs(D - T).max() <= (1e-12 if dtype == np.float64 else 2e-7) * scale
    # inverse: same spectrum through both
    Dc = T.astype(np.complex128)
    yi = omel.istft(Dc, HOP, dtype=np.float64)
    ti = torch.istft(torch.from_numpy(Dc), N_FFT, HOP, window=win, center=True).numpy()
    assert yi.shape == ti.shape == (HOP * 7,)
    assert np.abs(yi - ti).max() <= 1e-10 * np.abs(ti).max()
    assert np.abs(yi - y[: len(yi)].astype(np.float64)).max() <= (1e-10 if dtype == np.float64 else 1e-6)   # perfect reconstruction


@pytest.mark.parametrize("backend", BACKENDS)
def test_kernel_melspectrogram_agrees_with_the_torch_stft_route(backend):
    """HIP STFT + power + mel projection vs |torch.stft|^2 projected with a filterbank built HERE from the closed-form
    corners (float64 triangles, area normalised) — no product or oracle code on the expected side."""
    select(backend)
    from audiodiffusion.mel import Mel
    x_res = 8 if backend == "emu" else 64
    m = Mel(x_res=x_res, y_res=N_MELS, sample_rate=SR, n_fft=N_FFT, hop_length=HOP)
    corner = slaney_corner_hz(SR, N_MELS)
    hz = np.arange(N_FFT // 2 + 1) * SR / N_FFT
    fb = np.zeros((N_MELS, N_FFT // 2 + 1))
    for i in range(N_MELS):
        lo, mid, hi = corner[i], corner[i + 1], corner[i + 2]
        fb[i] = np.clip(np.minimum((hz - lo) / (mid - lo), (hi - hz) / (hi - mid)), 0, None) * 2.0 / (hi - lo)
    win = torch.hann_window(N_FFT, periodic=True, dtype=torch.float64)
    for dtype, tol in ((np.float32, 2e-5), (np.float64, 1e-10)):
        y = _noise_audio(m.slice_size, dtype, seed=3)
        got = m.audio_slices_to_melspectrograms([y])[0].astype(np.float64)
        S = torch.stft(torch.from_numpy(y.astype(np.float64)), N_FFT, HOP, window=win, center=True, pad_mode="constant",
                       return_complex=True).abs().numpy() ** 2
        want = fb @ S
        assert got.shape == want.shape == (N_MELS, x_res)
        assert np.abs(got - want).max() <= tol * want.max(), np.abs(got - want).max() / want.max()


# ------------------------------------------------------------------------------------------------ diffusion tables
def _closed_form_alphas(n=1000, b0=1e-4, b1=0.02):
    beta = np.array([b0 + (b1 - b0) * i / (n - 1) for i in range(n)], dtype=np.float64)
    return beta, np.exp(np.cumsum(np.log1p(-beta)))                 # alpha_bar_t = prod (1 - beta_i), via logs


@pytest.mark.parametrize("backend", BACKENDS)
def test_scheduler_tables_against_float64_closed_forms(backend):
    select(backend)
    from audiodiffusion import DDIMScheduler, DDPMScheduler
    beta, abar = _closed_form_alphas()
    for sched in (DDIMScheduler(), DDPMScheduler(), osched.DDIMScheduler(), osched.DDPMScheduler()):
        assert np.abs(sched.betas.double().numpy() - beta).max() <= 2e-9               # float32 ulp at 0.02 is 1.9e-9
        assert np.abs(sched.alphas_cumprod.double().numpy() / abar - 1).max() <= 5e-6   # 1000 float32 products
    # DDIM-50 "leading" spacing: t_i = (49 - i) * 20, previous timestep t - 20 (alpha_bar = 1 past the end)
    s = DDIMScheduler()
    s.set_timesteps(50)
    assert s.timesteps.tolist() == [(49 - i) * 20 for i in range(50)]
    for eta in (0.0, 1.0):
        for r, t in zip(s.coef_rows(eta), s.timesteps.tolist()):
            a_t, a_p = abar[t], (abar[t - 20] if t >= 20 else 1.0)
            sigma = eta * math.sqrt((1 - a_p) / (1 - a_t) * (1 - a_t / a_p))            # Song et al. 2021, eq. 16
            want = dict(sqrt_alpha=math.sqrt(a_t), sqrt_beta=math.sqrt(1 - a_t), k_x0=math.sqrt(a_p), k_x=0.0,
                        k_eps=math.sqrt(max(1 - a_p - sigma ** 2, 0.0)), k_noise=sigma, timestep=float(t), clip=1.0)
            for k, v in want.items():
                assert abs(r[k] - v) <= 2e-5 * max(1.0, abs(v)) + (2e-4 if k in ("k_eps", "k_noise") and t < 20 else 0), (eta, t, k, r[k], v)
    # DDPM-1000: posterior mean coefficients and "fixed_small" variance (Ho et al. 2020, eq. 7)
    p = DDPMScheduler()
    p.set_timesteps(1000)
    assert p.timesteps.tolist() == list(range(999, -1, -1))
    for r, t in zip(p.coef_rows(), p.timesteps.tolist()):
        a_t, a_p = abar[t], (abar[t - 1] if t >= 1 else 1.0)
        want = dict(sqrt_alpha=math.sqrt(a_t), sqrt_beta=math.sqrt(1 - a_t), k_x0=math.sqrt(a_p) * beta[t] / (1 - a_t),
                    k_x=math.sqrt(1 - beta[t]) * (1 - a_p) / (1 - a_t), k_eps=0.0,
                    k_noise=math.sqrt(max((1 - a_p) / (1 - a_t) * beta[t], 1e-20)) if t > 0 else 0.0, timestep=float(t))
        for k, v in want.items():
            # beta_t is recovered as 1 - abar_t/abar_{t-1} in float32 (as diffusers does): ~6e-8 absolute on 1e-4..2e-2
            assert abs(r[k] - v) <= 1.5e-3 * abs(v) + 1e-7, (t, k, r[k], v)
    # eta = 1 DDIM on the full 1000-step schedule IS the DDPM "fixed_small" sampler: same noise scale
    s.set_timesteps(1000)
    for rd, rp in zip(s.coef_rows(1.0), p.coef_rows()):
        assert abs(rd["k_noise"] - rp["k_noise"]) <= 2e-3 * max(rp["k_noise"], 1e-6) + 1e-7
