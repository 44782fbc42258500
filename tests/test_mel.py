"""Mel codec parity: HIP kernels (C-ABI adm_mel_forward / adm_mel_inverse) vs the numpy oracle restatement of
librosa (oracle/mel.py). Tolerances: filterbank tap indices bit-exact; u8 image identical in
>= 99.9 % of pixels and never more than 1 LSB apart; audio max|d| <= 1e-3*max|ref| with the same injected
Griffin-Lim phase; NNLS start point satisfies librosa's pgtol criterion."""
import numpy as np
import pytest
from PIL import Image

from native_backend import BACKENDS, select
from oracle import mel as omel


def _audio(n, seed=0, dtype=np.float32):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 22050
    return (0.1 * rng.standard_normal(n) + 0.5 * np.sin(2 * np.pi * 440 * t) + 0.3 * np.sin(2 * np.pi * 3000 * t)).astype(dtype)


SMALL = dict(x_res=32, y_res=64, hop_length=512, n_fft=1024, n_iter=3)


@pytest.mark.parametrize("backend", BACKENDS)
def test_filterbank_taps_bit_exact(backend):
    select(backend)
    from audiodiffusion.mel import Mel
    m = Mel()
    m._ensure_handle()
    start, count = m.filter_taps
    fb = omel.mel_filterbank(22050, 2048, 256)
    for i in range(256):
        nz = np.nonzero(fb[i] > 0)[0]
        assert start[i] == nz[0] and count[i] == nz[-1] - nz[0] + 1
    assert int(count.sum()) == 2032


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_audio_slice_to_image(backend, dtype):
    select(backend)
    from audiodiffusion.mel import Mel
    cfg = SMALL if backend == "emu" else {}
    mine, ref = Mel(**cfg), omel.Mel(**cfg)
    y = _audio(mine.slice_size * 2 + 100, dtype=dtype)
    for m in (mine, ref):
        m.load_audio(raw_audio=y)
    assert mine.get_number_of_slices() == ref.get_number_of_slices() == 2
    for s in (0, 1):
        a, b = np.asarray(mine.audio_slice_to_image(s)).astype(int), np.asarray(ref.audio_slice_to_image(s)).astype(int)
        assert a.shape == b.shape == (mine.y_res, mine.x_res)
        assert np.abs(a - b).max() <= 1
        assert (a == b).mean() >= 0.999, (a == b).mean()
    # silence -> all 255 (scripts/audio_to_images.py:46-48), short audio is zero-padded in float64 (mel.py:105)
    for m in (mine, ref):
        m.load_audio(raw_audio=np.zeros(100, dtype))
    assert (np.asarray(mine.audio_slice_to_image(0)) == 255).all()


@pytest.mark.parametrize("backend", BACKENDS)
def test_image_to_audio_with_injected_phase(backend):
    select(backend)
    from audiodiffusion.mel import Mel
    cfg = SMALL if backend == "emu" else {}
    mine, ref = Mel(**cfg), omel.Mel(**cfg)
    ref.load_audio(raw_audio=_audio(ref.slice_size + 1))
    img = ref.audio_slice_to_image(0)
    rng = np.random.default_rng(1)
    n_bins = 1 + mine.n_fft // 2
    phase = rng.random((n_bins, mine.x_res))
    info = []
    ref_mag = ref.image_to_stft_magnitude(img, info)
    assert all(d["nit"] == 0 for d in info)
    ref_audio = omel.griffinlim(ref_mag, ref.n_iter, ref.hop_length, ref.n_fft, init_phase=phase)
    audio, mag = mine.images_to_audios([img], init_phase=phase[None], return_magnitude=True)
    assert mine.last_nnls_pg is not None and mine.last_nnls_pg <= 1e-5
    # compared as POWER: untouched rows (DC, Nyquist) hold pseudo-inverse rounding noise around 0 (see tests/test_golden.py)
    assert np.abs(mag[0] ** 2 - ref_mag ** 2).max() <= 1e-12 * max(1.0, np.abs(ref_mag ** 2).max())
    assert audio.shape == (1, mine.hop_length * (mine.x_res - 1)) and audio.dtype == np.float32
    err = np.abs(audio[0] - ref_audio).max() / np.abs(ref_audio).max()
    assert err <= 1e-3, err
    # the reference call form: unseeded phase, result differs run to run but has the right length/dtype
    a2 = mine.image_to_audio(img)
    assert a2.shape == ref_audio.shape and np.isfinite(a2).all()


@pytest.mark.parametrize("backend", BACKENDS)
def test_batched_forward_matches_single(backend):
    select(backend)
    from audiodiffusion.mel import Mel
    m = Mel(**SMALL)
    ys = [_audio(m.slice_size, seed=s) for s in range(3)]
    batch = m.audio_slices_to_images(ys)
    for i, y in enumerate(ys):
        m.load_audio(raw_audio=y)
        assert np.array_equal(batch[i], np.asarray(m.audio_slice_to_image(0)))


NNLS_ITERATING = [  # (sample_rate, cfg): low rates make the Slaney triangles tall enough that L-BFGS-B does NOT stop at its start point
    (1000, dict(x_res=8, y_res=8, n_fft=128, hop_length=32)),
    (200, dict(x_res=4, y_res=16, n_fft=256, hop_length=64)),
    (50, dict(x_res=4, y_res=16, n_fft=256, hop_length=64)),
    (1000, dict(x_res=16, y_res=4, n_fft=64, hop_length=16)),
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", NNLS_ITERATING, ids=[f"sr{c[0]}-{c[1]['y_res']}mels" for c in NNLS_ITERATING])
def test_nnls_solver_where_lbfgsb_iterates(backend, case):
    """M7 (`mel.py:165` -> librosa.util.nnls): where scipy's L-BFGS-B really iterates (nit > 0) the device solver must reach
    an objective at least as low (f_hip <= f_oracle * (1 + 1e-3)) and satisfy the same stopping rule."""
    select(backend)
    from audiodiffusion.mel import Mel
    sr, cfg = case
    mine, ref = Mel(sample_rate=sr, n_iter=1, **cfg), omel.Mel(sample_rate=sr, n_iter=1, **cfg)
    rng = np.random.default_rng(sr)
    img = Image.fromarray(rng.integers(0, 256, (cfg["y_res"], cfg["x_res"]), dtype=np.uint8))
    info = []
    ref_mag = ref.image_to_stft_magnitude(img, info)
    assert all(d["nit"] > 0 for d in info), info                      # the regime this test is about
    phase = rng.random((1, 1 + cfg["n_fft"] // 2, cfg["x_res"]))
    _, mag = mine.images_to_audios([img], init_phase=phase, return_magnitude=True)
    S = omel.db_to_power(np.asarray(img).astype(float) * mine.top_db / 255 - mine.top_db)
    A = omel.mel_filterbank(sr, cfg["n_fft"], cfg["y_res"], np.float64)
    f = lambda X: 0.5 * np.sum((A @ X - S) ** 2) / S.size  # noqa: E731  (librosa's objective, _nnls_obj)
    f_hip, f_ref, f_start = f(mag[0] ** 2), f(ref_mag ** 2), f(np.clip(np.linalg.pinv(A) @ S, 0, None))
    assert mine.last_nnls_pg_start > 1e-5 and mine.last_nnls_iterations > 0
    assert mine.last_nnls_pg <= 1e-5                                  # L-BFGS-B's own stopping rule holds at the result
    assert (mag[0] >= 0).all()
    assert f_hip <= f_ref * (1 + 1e-3), (f_hip, f_ref)
    assert f_hip < f_start


@pytest.mark.parametrize("backend", BACKENDS)
def test_nnls_solver_leaves_converged_blocks_alone(backend):
    """The usual regime: the start point already satisfies pgtol, scipy returns it after 0 iterations — so must the device."""
    select(backend)
    from audiodiffusion.mel import Mel
    cfg = dict(x_res=16, y_res=16, n_fft=256, hop_length=64, n_iter=1)
    mine, ref = Mel(**cfg), omel.Mel(**cfg)
    rng = np.random.default_rng(5)
    img = Image.fromarray(rng.integers(0, 256, (16, 16), dtype=np.uint8))
    info = []
    ref_mag = ref.image_to_stft_magnitude(img, info)
    assert all(d["nit"] == 0 for d in info)
    _, mag = mine.images_to_audios([img], init_phase=rng.random((1, 129, 16)), return_magnitude=True)
    assert mine.last_nnls_iterations == 0 and mine.last_nnls_pg == mine.last_nnls_pg_start <= 1e-5
    assert np.abs(mag[0] ** 2 - ref_mag ** 2).max() <= 1e-12 * max(1.0, np.abs(ref_mag ** 2).max())


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fast_path_n_fft_2048_image_matches_the_oracle(backend, dtype):
    """n_fft = 2048 takes the wave-per-frame radix-16 real-FFT kernel and the grid dB kernel (per-spectrogram maximum from the
    STFT kernel's atomics): same bars as the generic path, on a frame count that leaves the last workgroup ragged."""
    select(backend)
    from audiodiffusion.mel import Mel
    cfg = dict(x_res=11, y_res=256, n_fft=2048, hop_length=512)      # 11 frames: workgroups of 8 frames, the second one ragged
    mine, ref = Mel(**cfg), omel.Mel(**cfg)
    y = _audio(mine.slice_size * 2 + 7, seed=4, dtype=dtype)
    for m in (mine, ref):
        m.load_audio(raw_audio=y)
    got = mine.audio_slices_to_images([mine.get_audio_slice(0), mine.get_audio_slice(1)])
    for s in (0, 1):
        a, b = got[s].astype(int), np.asarray(ref.audio_slice_to_image(s)).astype(int)
        assert a.shape == b.shape == (256, 11)
        assert np.abs(a - b).max() <= 1 and (a == b).mean() >= 0.999, (np.abs(a - b).max(), (a == b).mean())


@pytest.mark.parametrize("backend", BACKENDS)
def test_fast_path_n_fft_2048_griffin_lim_matches_the_oracle(backend):
    """Griffin-Lim with n_fft = 2048: the inverse real FFT (half spectrum -> frame) and the forward real FFT of the update
    step both run on the wave-per-frame radix-16 engine; 6 frames leave the second workgroup (4 frames each) ragged."""
    select(backend)
    from audiodiffusion.mel import Mel
    cfg = dict(x_res=6, y_res=32, n_fft=2048, hop_length=512, n_iter=4)
    mine, ref = Mel(**cfg), omel.Mel(**cfg)
    ref.load_audio(raw_audio=_audio(ref.slice_size + 1, seed=6))
    img = ref.audio_slice_to_image(0)
    phase = np.random.default_rng(3).random((2, 1025, 6))
    want = np.stack([ref.image_to_audio(img, init_phase=phase[i]) for i in range(2)])
    got = mine.images_to_audios([img, img], init_phase=phase)
    assert got.shape == want.shape == (2, 512 * 5)
    assert np.abs(got - want).max() <= 1e-3 * np.abs(want).max(), np.abs(got - want).max() / np.abs(want).max()
