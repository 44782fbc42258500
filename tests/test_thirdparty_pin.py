"""Mel forward path pinned to THIRD-PARTY vectors: tests/golden/thirdparty_mel.npz holds the outputs of
`transformers.audio_utils` (mel_filter_bank / spectrogram / power_to_db; see tests/golden/make_thirdparty_mel.py) — numbers
that neither oracle/ nor the HIP kernels produced.  (a) the oracle reproduces them (CPU); (b) the product path — sparse
filterbank generator, wave-per-frame FFT kernel, dB + u8 kernel — reproduces them (emulator here, MI355X with -m gpu)."""
import os
import sys

import numpy as np
import pytest

from native_backend import BACKENDS, select

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, G)
from make_thirdparty_mel import CONFIGS, clips, digest  # noqa: E402

Z = np.load(os.path.join(G, "thirdparty_mel.npz"))


def _dense_fb(name, c):
    fb = np.zeros((c["y_res"], c["n_fft"] // 2 + 1))
    fb[Z[f"{name}:fb_rows"], Z[f"{name}:fb_cols"]] = Z[f"{name}:fb_vals"]
    return fb


def _audio(name, c):
    ys = clips(c)
    for i, y in enumerate(ys):
        assert digest(y) == str(Z[f"{name}:audio_sha256_{i}"]), "regenerated test audio differs from the fixture's input"
    return ys


def _mel_kwargs(c):
    return dict(x_res=c["x_res"], y_res=c["y_res"], sample_rate=c["sample_rate"], n_fft=c["n_fft"],
                hop_length=c["hop_length"], top_db=c["top_db"])


def _check_images(img, name, i):
    want = Z[f"{name}:image{i}"]
    got = np.asarray(img)
    assert got.shape == want.shape and got.dtype == np.uint8
    # the third-party implementation runs the whole chain in float64, librosa (and so the oracle and the kernels) keeps
    # float32 audio in float32 after the FFT: dB values differ by ~1e-5, which moves a quantisation tie now and then
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
    assert (got == want).mean() >= 0.999


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_oracle_matches_the_thirdparty_vectors(name):
    from oracle import mel as omel
    c = CONFIGS[name]
    fb3 = _dense_fb(name, c)
    fb64 = omel.mel_filterbank(c["sample_rate"], c["n_fft"], c["y_res"], np.float64)
    assert np.array_equal(fb64 != 0, fb3 != 0)                      # the bin-indexing claim, against foreign numbers
    assert np.abs(fb64 - fb3).max() <= 1e-12 * fb3.max()   # float64 rounding of two different evaluation orders
    fb32 = omel.mel_filterbank(c["sample_rate"], c["n_fft"], c["y_res"], np.float32)
    assert np.abs(fb32 - fb3).max() <= 2e-7 * fb3.max()
    ys = _audio(name, c)
    S = omel.melspectrogram(ys[0], c["sample_rate"], c["n_fft"], c["hop_length"], c["y_res"])
    P = Z[f"{name}:power0"]
    assert np.abs(S - P).max() <= 2e-6 * P.max() and (np.abs(S - P) / np.maximum(P, 1e-9 * P.max())).max() <= 1e-5
    for i, y in enumerate(ys):
        m = omel.Mel(**_mel_kwargs(c))
        m.load_audio(raw_audio=y)
        _check_images(m.audio_slice_to_image(0), name, i)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_product_path_matches_the_thirdparty_vectors(backend, name):
    select(backend)
    from audiodiffusion import Mel
    from audiodiffusion.mel import slaney_filter_taps, taps_to_dense
    c = CONFIGS[name]
    fb3 = _dense_fb(name, c)
    start, count, w32, w64 = slaney_filter_taps(c["sample_rate"], c["n_fft"], c["y_res"])
    dense = taps_to_dense(start, count, w64, c["n_fft"] // 2 + 1)
    assert np.array_equal(dense != 0, fb3 != 0) and np.abs(dense - fb3).max() <= 1e-12 * fb3.max()
    ys = _audio(name, c)
    m = Mel(**_mel_kwargs(c))
    S = m.audio_slices_to_melspectrograms([ys[0]])[0]
    P = Z[f"{name}:power0"]
    assert np.abs(S - P).max() <= 2e-6 * P.max() and (np.abs(S - P) / np.maximum(P, 1e-9 * P.max())).max() <= 1e-5
    for i, y in enumerate(ys):
        m.load_audio(raw_audio=y)
        _check_images(m.audio_slice_to_image(0), name, i)
