"""Oracle restatement of `audiodiffusion/mel.py:44-168` and of the librosa==0.10.2.post1
routines it calls (TEST INFRASTRUCTURE ONLY).

librosa is not vendored in the reference and not installed here; the functions
below restate its published algorithms (core/spectrum.py stft/istft/griffinlim/
power_to_db/db_to_power, feature/spectral.py melspectrogram, feature/inverse.py
mel_to_stft/mel_to_audio, filters.py mel, util/_nnls.py) including their dtype
behaviour. Parity unpinned (see oracle/__init__.py).

The only deliberate extension: `image_to_audio(..., init_phase=)` lets a test inject
the Griffin-Lim initial phase, which the reference draws unseeded (mel.py:165-167).
"""
import numpy as np
import scipy.optimize
import scipy.signal
from PIL import Image

# --------------------------------------------------------------------------- librosa.filters


def hz_to_mel(f):
    f = np.asanyarray(f, dtype=float)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if f.ndim:
        m = f >= min_log_hz
        mels[m] = min_log_mel + np.log(f[m] / min_log_hz) / logstep
    elif f >= min_log_hz:
        mels = min_log_mel + np.log(f / min_log_hz) / logstep
    return mels


def mel_to_hz(mels):
    mels = np.asanyarray(mels, dtype=float)
    f_sp = 200.0 / 3
    freqs = f_sp * mels
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if mels.ndim:
        m = mels >= min_log_mel
        freqs[m] = min_log_hz * np.exp(logstep * (mels[m] - min_log_mel))
    elif mels >= min_log_mel:
        freqs = min_log_hz * np.exp(logstep * (mels - min_log_mel))
    return freqs


def mel_filterbank(sr, n_fft, n_mels, dtype=np.float32):
    """librosa.filters.mel(htk=False, norm="slaney", fmin=0, fmax=sr/2)."""
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=dtype)
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


# --------------------------------------------------------------------------- librosa.core.spectrum


def _hann(n_fft):
    return scipy.signal.get_window("hann", n_fft, fftbins=True)  # periodic, float64


def _dtype_r2c(d):
    return {np.dtype(np.float32): np.complex64, np.dtype(np.float64): np.complex128}.get(np.dtype(d), np.complex64)


def stft(y, n_fft, hop_length):
    """center=True, pad_mode="constant", window="hann", win_length=n_fft.
    The window is float64 so the FFT runs in double and is rounded to the r2c dtype of y."""
    win = _hann(n_fft).reshape(-1, 1)
    yp = np.pad(y, n_fft // 2, mode="constant")
    n_frames = 1 + (len(yp) - n_fft) // hop_length
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(n_frames)[None, :]
    frames = yp[idx]
    out = np.empty((1 + n_fft // 2, n_frames), dtype=_dtype_r2c(y.dtype))
    out[:] = np.fft.rfft(win * frames, axis=0)
    return out


def window_sumsquare(n_frames, hop_length, n_fft, dtype=np.float32):
    n = n_fft + hop_length * (n_frames - 1)
    x = np.zeros(n, dtype=dtype)
    win_sq = _hann(n_fft) ** 2
    for i in range(n_frames):
        s = i * hop_length
        x[s : min(n, s + n_fft)] += win_sq[: max(0, min(n_fft, n - s))]
    return x


def istft(D, hop_length, dtype=np.float32):
    """center=True, length=None; overlap-add in `dtype`, then window-sum-square normalisation."""
    n_fft = 2 * (D.shape[0] - 1)
    n_frames = D.shape[1]
    win = _hann(n_fft).reshape(-1, 1)
    full = n_fft + hop_length * (n_frames - 1)
    y = np.zeros(full, dtype=dtype)
    ytmp = win * np.fft.irfft(D, n=n_fft, axis=0)
    for f in range(n_frames):
        s = f * hop_length
        y[s : s + n_fft] += ytmp[:, f]
    y = y[n_fft // 2 : full - n_fft // 2]
    wss = window_sumsquare(n_frames, hop_length, n_fft, dtype=dtype)[n_fft // 2 :]
    wss = wss[: len(y)]
    nz = wss > np.finfo(dtype).tiny
    y[nz] /= wss[nz]
    return y


def power_to_db(S, ref=np.max, amin=1e-10, top_db=80.0):
    magnitude = S
    ref_value = ref(magnitude) if callable(ref) else np.abs(ref)
    log_spec = 10.0 * np.log10(np.maximum(amin, magnitude))
    log_spec -= 10.0 * np.log10(np.maximum(amin, ref_value))
    if top_db is not None:
        log_spec = np.maximum(log_spec, log_spec.max() - top_db)
    return log_spec


def db_to_power(S_db, ref=1.0):
    return ref * np.power(10.0, 0.1 * S_db)


def melspectrogram(y, sr, n_fft, hop_length, n_mels):
    S = np.abs(stft(y, n_fft, hop_length)) ** 2.0
    basis = mel_filterbank(sr, n_fft, n_mels)  # float32
    return np.einsum("ft,mf->mt", S, basis, optimize=True)


# --------------------------------------------------------------------------- librosa.util.nnls / feature.inverse


def _nnls_obj(x, shape, A, B):
    x = x.reshape(shape)
    diff = np.einsum("mf,ft->mt", A, x, optimize=True) - B
    value = (1 / B.size) * 0.5 * np.sum(diff**2)
    grad = (1 / B.size) * np.einsum("mf,mt->ft", A, diff, optimize=True)
    return value, grad.flatten()


def _nnls_lbfgs_block(A, B, x_init, info=None):
    shape = x_init.shape
    bounds = [(0, None)] * x_init.size
    x, obj, d = scipy.optimize.fmin_l_bfgs_b(_nnls_obj, x_init, args=(shape, A, B), bounds=bounds, m=A.shape[1])
    if info is not None:
        info.append(dict(obj=obj, nit=d["nit"], funcalls=d["funcalls"], warnflag=d["warnflag"]))
    return x.reshape(shape)


def nnls(A, B, info=None, lbfgs=True):
    """librosa.util.nnls for 2-D B: pinv initial guess clipped at 0, then blockwise L-BFGS-B
    (block = 2**18 // (B.shape[0] * itemsize) columns; `m` = A.shape[1])."""
    n_columns = max((2**8 * 2**10) // (int(np.prod(B.shape[:-1])) * A.itemsize), 1)
    x = np.einsum("fm,mt->ft", np.linalg.pinv(A), B, optimize=True)
    np.clip(x, 0, None, out=x)
    if not lbfgs:
        return x
    if B.shape[-1] <= n_columns:
        return _nnls_lbfgs_block(A, B, x, info).astype(A.dtype)
    x_init = x
    for s in range(0, x.shape[-1], n_columns):
        t = min(s + n_columns, B.shape[-1])
        x[:, s:t] = _nnls_lbfgs_block(A, B[:, s:t], x_init[:, s:t], info)
    return x


def mel_to_stft(M, sr, n_fft, power=2.0, info=None, lbfgs=True):
    basis = mel_filterbank(sr, n_fft, M.shape[-2], dtype=M.dtype)
    inverse = nnls(basis, M, info, lbfgs)
    return np.power(inverse, 1.0 / power, out=inverse)


def griffinlim(S, n_iter, hop_length, n_fft, momentum=0.99, init_phase=None, rng=None, dtype=np.float32):
    """init="random": angles = exp(2j*pi*U[0,1)); `init_phase` (same shape as S, radians/(2*pi) in [0,1))
    replaces the unseeded draw."""
    angles = np.empty(S.shape, dtype=_dtype_r2c(S.dtype))
    eps = np.finfo(angles.real.dtype).tiny
    if init_phase is None:
        rng = rng or np.random.default_rng()
        init_phase = rng.random(size=S.shape)
    ph = 2 * np.pi * init_phase
    angles[:] = np.cos(ph) + 1j * np.sin(ph)
    rebuilt = None
    tprev = None
    angles *= S
    for _ in range(n_iter):
        inverse = istft(angles, hop_length, dtype=dtype)
        rebuilt = stft(inverse, n_fft, hop_length)
        angles[:] = rebuilt
        if tprev is not None:
            angles -= (momentum / (1 + momentum)) * tprev
        angles /= np.abs(angles) + eps
        angles *= S
        rebuilt, tprev = tprev, rebuilt
    return istft(angles, hop_length, dtype=dtype)


# --------------------------------------------------------------------------- audiodiffusion/mel.py


class Mel:
    """`audiodiffusion/mel.py:44-168` restated line by line."""

    def __init__(self, x_res=256, y_res=256, sample_rate=22050, n_fft=2048, hop_length=512, top_db=80, n_iter=32):
        self.hop_length = hop_length
        self.sr = sample_rate
        self.n_fft = n_fft
        self.top_db = top_db
        self.n_iter = n_iter
        self.set_resolution(x_res, y_res)
        self.audio = None

    def set_resolution(self, x_res, y_res):  # mel.py:80-90
        self.x_res = x_res
        self.y_res = y_res
        self.n_mels = self.y_res
        self.slice_size = self.x_res * self.hop_length - 1

    def load_audio(self, audio_file=None, raw_audio=None):  # mel.py:92-106 (file decode out of scope)
        assert audio_file is None, "oracle takes raw_audio only"
        self.audio = raw_audio
        if len(self.audio) < self.x_res * self.hop_length:
            self.audio = np.concatenate([self.audio, np.zeros((self.x_res * self.hop_length - len(self.audio),))])

    def get_number_of_slices(self):
        return len(self.audio) // self.slice_size

    def get_audio_slice(self, slice=0):
        return self.audio[self.slice_size * slice : self.slice_size * (slice + 1)]

    def get_sample_rate(self):
        return self.sr

    def audio_slice_to_image(self, slice, ref=np.max):  # mel.py:135-151
        S = melspectrogram(self.get_audio_slice(slice), self.sr, self.n_fft, self.hop_length, self.n_mels)
        log_S = power_to_db(S, ref=ref, top_db=self.top_db)
        bytedata = (((log_S + self.top_db) * 255 / self.top_db).clip(0, 255) + 0.5).astype(np.uint8)
        return Image.fromarray(bytedata)

    def image_to_stft_magnitude(self, image, info=None, lbfgs=True):  # mel.py:162-164 + mel_to_stft
        bytedata = np.frombuffer(image.tobytes(), dtype="uint8").reshape((image.height, image.width))
        log_S = bytedata.astype("float") * self.top_db / 255 - self.top_db
        S = db_to_power(log_S)
        return mel_to_stft(S, self.sr, self.n_fft, 2.0, info, lbfgs)

    def image_to_audio(self, image, init_phase=None, info=None, lbfgs=True):  # mel.py:153-168
        mag = self.image_to_stft_magnitude(image, info, lbfgs)
        return griffinlim(mag, self.n_iter, self.hop_length, self.n_fft, init_phase=init_phase)
