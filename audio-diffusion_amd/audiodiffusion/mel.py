"""Mel drop-in (`audiodiffusion/mel.py:44-168` of the reference): same constructor, attributes, methods and
`mel_config.json` serialisation; the arithmetic (`librosa` in the reference) runs in HIP kernels (csrc/k_mel.hip).

Host side (this file) only prepares the constant tables a `Mel` configuration implies — the periodic Hann window
(scipy, as librosa does), FFT twiddles, the Slaney mel filterbank in CSR/CSC form, its fp64 pseudo-inverse and the
window-sum-square envelope — and moves audio/images between numpy/PIL and device memory.

Beyond the reference API, `audio_slices_to_images` / `images_to_audios` convert whole batches in one launch
sequence (the reference loops serially over images on one host core, `pipeline_audio_diffusion.py:201`), and
`image_to_audio(..., init_phase=)` lets a caller inject the Griffin-Lim start phase, which librosa draws unseeded.
File decoding (`librosa.load`, mel.py:100) is out of scope: WAV files are read with scipy, other formats raise.
"""
import ctypes as C
import json
import os
import warnings
from typing import Callable, Union

import numpy as np
import torch
from PIL import Image

from . import _native as N

class MelConfigStruct(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("x_res", "y_res", "sample_rate", "n_fft", "hop_length", "top_db", "n_iter")]


# ---- constant tables (host, fp64) -------------------------------------------------------------------------------

def _slaney_hz(mel):
    """Slaney's auditory scale, mel -> Hz: linear (200/3 Hz per mel) below 1 kHz = 15 mel, then 27 log-spaced steps per
    factor 6.4 (librosa.mel_to_hz, htk=False)."""
    mel = np.asarray(mel, dtype=np.float64)
    lin = (200.0 / 3.0) * mel
    k = 1000.0 / (200.0 / 3.0)                            # the 1 kHz knee in mel: 15 up to one ulp, as librosa computes it
    return np.where(mel >= k, 1000.0 * np.exp((np.log(6.4) / 27.0) * (mel - k)), lin)


def _slaney_mel(hz):
    hz = float(hz)
    return 1000.0 / (200.0 / 3.0) + np.log(hz / 1000.0) / (np.log(6.4) / 27.0) if hz >= 1000.0 else hz / (200.0 / 3.0)


def slaney_filter_taps(sr, n_fft, n_mels):
    """The Slaney-normalised triangular filterbank librosa.filters.mel(sr, n_fft, n_mels) builds (htk=False, fmin=0,
    fmax=sr/2), generated SPARSE, filter by filter, from the triangle's corner frequencies instead of a dense
    (n_mels, n_bins) matrix: filter m rises from lo = edge[m] to its apex edge[m+1] and falls to hi = edge[m+2]; its
    non-zero taps are the FFT bins strictly inside (lo, hi). Returns (start, count, w32, w64): first bin and number of
    bins per filter and the concatenated weights as librosa stores them for float32 audio (triangle rounded to float32,
    THEN scaled by the float64 area normalisation 2/(hi-lo) and rounded again) and for float64 audio."""
    n_bins = 1 + n_fft // 2
    edge = _slaney_hz(np.linspace(_slaney_mel(0.0), _slaney_mel(sr / 2.0), n_mels + 2))
    bin_hz = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    start, count = np.zeros(n_mels, np.int32), np.zeros(n_mels, np.int32)
    w32, w64 = [], []
    for m in range(n_mels):
        lo, apex, hi = edge[m], edge[m + 1], edge[m + 2]
        first = int(np.searchsorted(bin_hz, lo, side="right"))          # first bin with f > lo
        last = int(np.searchsorted(bin_hz, hi, side="left")) - 1        # last bin with f < hi
        f = bin_hz[first:last + 1]
        tri = np.minimum((f - lo) / (apex - lo), (hi - f) / (hi - apex))
        keep = np.nonzero(tri > 0)[0]                                   # guards against a bin rounding onto a corner
        if len(keep) == 0:
            continue
        first, f, tri = first + int(keep[0]), f[keep[0]:keep[-1] + 1], tri[keep[0]:keep[-1] + 1]
        area = 2.0 / (hi - lo)
        start[m], count[m] = first, len(f)
        w64.append(tri * area)
        w32.append((tri.astype(np.float32).astype(np.float64) * area).astype(np.float32))
    w32 = np.concatenate(w32).astype(np.float32) if w32 else np.zeros(0, np.float32)
    w64 = np.concatenate(w64).astype(np.float64) if w64 else np.zeros(0, np.float64)
    return start, count, w32, w64


def taps_to_dense(start, count, w, n_bins):
    out = np.zeros((len(start), n_bins), dtype=w.dtype)
    o = 0
    for m, (s0, c) in enumerate(zip(start, count)):
        out[m, s0:s0 + c] = w[o:o + c]
        o += c
    return out


def _window_sumsquare(window, n_frames, hop, n_fft):
    x = np.zeros(n_fft + hop * (n_frames - 1), dtype=np.float32)  # float32 accumulation, as librosa's istft
    win_sq = window**2
    for i in range(n_frames):
        s = i * hop
        x[s : s + n_fft] += win_sq[: max(0, min(n_fft, len(x) - s))]
    return x


class Mel:
    """
    Parameters (as the reference, `audiodiffusion/mel.py:45-68`):
        x_res (`int`): x resolution of spectrogram (time)
        y_res (`int`): y resolution of spectrogram (frequency bins)
        sample_rate (`int`): sample rate of audio
        n_fft (`int`): number of Fast Fourier Transforms
        hop_length (`int`): hop length (a higher number is recommended for lower than 256 y_res)
        top_db (`int`): loudest in decibels
        n_iter (`int`): number of iterations for Griffin Linn mel inversion
    """

    config_name = "mel_config.json"

    def __init__(self, x_res: int = 256, y_res: int = 256, sample_rate: int = 22050, n_fft: int = 2048,
                 hop_length: int = 512, top_db: int = 80, n_iter: int = 32):
        self.hop_length = hop_length
        self.sr = sample_rate
        self.n_fft = n_fft
        self.top_db = top_db
        self.n_iter = n_iter
        self.set_resolution(x_res, y_res)
        self.audio = None
        self.last_nnls_pg = None       # max |projected gradient| of the NNLS solution the last image_to_audio returned
        self.last_nnls_pg_start = None  # ... of its start point clip(pinv(A) S, 0)
        self.last_nnls_iterations = None  # most solver iterations any column needed (0: start point returned, scipy nit = 0)
        self._nnls_lip = None
        self._nnls_max_iter = 4000

    @property
    def nnls_max_iter(self):
        """Iteration cap of the device NNLS solver (the column blocks where librosa's L-BFGS-B would iterate). Setting it
        reaches the native handle at once. Those blocks are OUTSIDE sample parity with librosa: the minimiser of the
        underdetermined problem is not unique — only the stopping rule (projected gradient <= pgtol) and f <= f_scipy hold."""
        return self._nnls_max_iter

    @nnls_max_iter.setter
    def nnls_max_iter(self, v):
        self._nnls_max_iter = int(v)
        if getattr(self, "_handle", None) is not None:
            N.check(N.lib().adm_mel_set_nnls_solver(self._handle, self._nnls_lip, self._nnls_max_iter))

    @property
    def config(self):
        return dict(x_res=self.x_res, y_res=self.y_res, sample_rate=self.sr, n_fft=self.n_fft,
                    hop_length=self.hop_length, top_db=self.top_db, n_iter=self.n_iter)

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
        d = {"_class_name": "Mel", "_diffusers_version": "0.24.0"}
        d.update(self.config)
        with open(os.path.join(path, self.config_name), "w") as f:
            json.dump(d, f, indent=2, sort_keys=True)

    def set_resolution(self, x_res: int, y_res: int):
        self.x_res = x_res
        self.y_res = y_res
        self.n_mels = self.y_res
        self.slice_size = self.x_res * self.hop_length - 1
        self._free()

    # ---- native handle -------------------------------------------------------------------------------------
    _handle = None

    def _free(self):
        if getattr(self, "_handle", None) is not None:
            N.lib().adm_mel_destroy(self._handle)
        self._handle = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass

    def _device(self):
        return N.default_device()

    def _ensure_handle(self):
        if self._handle is not None:
            return self._handle
        import scipy.signal
        lib = N.lib()
        if not hasattr(lib, "adm_mel_create"):
            raise N.NativeError("the native library was built without csrc/k_mel.hip")
        n_fft, n_bins = self.n_fft, 1 + self.n_fft // 2
        window = scipy.signal.get_window("hann", n_fft, fftbins=True).astype(np.float64)
        q = np.arange(n_fft // 2)
        tw = np.stack([np.cos(2 * np.pi * q / n_fft), -np.sin(2 * np.pi * q / n_fft)], axis=1).astype(np.float64)
        start, count, w32, w64 = slaney_filter_taps(self.sr, n_fft, self.n_mels)
        fb64 = taps_to_dense(start, count, w64, n_bins)
        # librosa's dense matrix carries a NEGATIVE zero where an FFT bin sits exactly on a filter's lower corner (bin 0 on
        # filter 0: -(0 - 0) / df = -0.0 survives its np.maximum(0, .)). The value is the same, but LAPACK's SVD inside pinv
        # propagates the sign into the rounding noise of the rows no filter touches (DC), and 32 Griffin-Lim iterations
        # amplify that to 1e-3 of the audio: keep the bit pattern librosa has.
        edge0 = _slaney_hz(np.linspace(_slaney_mel(0.0), _slaney_mel(self.sr / 2.0), self.n_mels + 2))
        on_corner = np.fft.rfftfreq(n=n_fft, d=1.0 / self.sr)[None, :] == edge0[:self.n_mels, None]
        fb64[on_corner] = -0.0
        self.filter_taps = (start.copy(), count.copy())
        # CSC (per FFT bin: the mel filters touching it), same tap set
        t_off = np.zeros(n_bins + 1, np.int32)
        t_idx, t_w = [], []
        for f in range(n_bins):
            ms = [m for m in range(self.n_mels) if start[m] <= f < start[m] + count[m]]
            t_off[f + 1] = t_off[f] + len(ms)
            t_idx += ms
            t_w += [fb64[m, f] for m in ms]
        t_idx = np.asarray(t_idx, np.int32)
        t_w = np.asarray(t_w, np.float64)
        assert len(t_idx) == len(w64)
        pinv = np.ascontiguousarray(np.linalg.pinv(fb64))  # (n_bins, n_mels), librosa.util.nnls start point
        wss = _window_sumsquare(window, self.x_res, self.hop_length, n_fft)[n_fft // 2 :][: self.hop_length * (self.x_res - 1)]
        wss = np.ascontiguousarray(wss, np.float32)
        nnls_cols = max((2**8 * 2**10) // (self.n_mels * 8), 1)  # librosa MAX_MEM_BLOCK column blocking
        cfg = MelConfigStruct(self.x_res, self.y_res, self.sr, n_fft, self.hop_length, int(self.top_db), self.n_iter)
        h = C.c_void_p()
        keep = [window, tw, start, count, w32, w64, t_off, t_idx, t_w, pinv, wss]
        args = [a.ctypes.data_as(C.c_void_p) for a in keep]
        N.check(lib.adm_mel_create(C.byref(cfg), args[0], args[1], args[2], args[3], args[4], args[5], int(len(w64)),
                                   args[6], args[7], args[8], args[9], args[10], nnls_cols, C.byref(h)))
        # NNLS solver for the blocks L-BFGS-B would iterate on: step 1 / lambda_max(A A^T)
        self._nnls_lip = float(np.linalg.eigvalsh(fb64 @ fb64.T)[-1])
        N.check(lib.adm_mel_set_nnls_solver(h, self._nnls_lip, self._nnls_max_iter))
        self._handle = h
        return h

    # ---- reference API -------------------------------------------------------------------------------------
    def load_audio(self, audio_file: str = None, raw_audio: np.ndarray = None):
        """Load audio (`mel.py:92-106`). Files: WAV via scipy (other rates are resampled with scipy's polyphase filter,
        not librosa's soxr_hq: decoding is outside the parity scope)."""
        if audio_file is not None:
            import scipy.io.wavfile
            sr, data = scipy.io.wavfile.read(audio_file)
            if data.dtype.kind == "u":                    # 8-bit WAV is offset binary
                data = (data.astype(np.float32) - 128.0) / 128.0
            elif data.dtype.kind == "i":                  # libsndfile's (= librosa.load's) normalisation: / 2^(bits-1)
                data = data.astype(np.float32) / float(-int(np.iinfo(data.dtype).min))
            if data.ndim > 1:
                data = data.mean(axis=1)
            if sr != self.sr:
                # librosa.load resamples with soxr_hq (`mel.py:100`); soxr is not available here, so a file at another
                # rate goes through scipy's polyphase (Kaiser-windowed) resampler: band-limited and length-consistent, but
                # NOT sample-identical to the reference — file decoding is outside the parity scope
                import math
                import scipy.signal
                g = math.gcd(int(sr), int(self.sr))
                data = scipy.signal.resample_poly(data.astype(np.float64), self.sr // g, sr // g)
            self.audio = data.astype(np.float32)
        else:
            self.audio = raw_audio
        # Pad with silence if necessary.
        if len(self.audio) < self.x_res * self.hop_length:
            self.audio = np.concatenate([self.audio, np.zeros((self.x_res * self.hop_length - len(self.audio),))])

    def get_number_of_slices(self) -> int:
        return len(self.audio) // self.slice_size

    def get_audio_slice(self, slice: int = 0) -> np.ndarray:
        return self.audio[self.slice_size * slice : self.slice_size * (slice + 1)]

    def get_sample_rate(self) -> int:
        return self.sr

    def audio_slices_to_images(self, slices) -> np.ndarray:
        """Batched `audio_slice_to_image`: list of equal-length 1-D arrays (or a 2-D array) -> (B, y_res, frames) uint8."""
        arr = np.stack([np.asarray(s) for s in slices])
        if arr.dtype not in (np.float32, np.float64):
            arr = arr.astype(np.float32 if arr.dtype.itemsize <= 4 else np.float64)
        h = self._ensure_handle()
        dev = self._device()
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
        B, n = t.shape
        frames = 1 + n // self.hop_length
        out = torch.empty((B, self.n_mels, frames), dtype=torch.uint8, device=dev)
        N.check(N.lib().adm_mel_forward(h, N.ptr(t), int(arr.dtype == np.float64), B, n, n, N.ptr(out), N.stream_for(t)))
        return out.cpu().numpy()

    def audio_slices_to_melspectrograms(self, slices) -> np.ndarray:
        """librosa.feature.melspectrogram of every slice (`mel.py:140-147`, before the dB conversion):
        (B, y_res, frames) in the slices' precision."""
        arr = np.stack([np.asarray(s) for s in slices])
        if arr.dtype not in (np.float32, np.float64):
            arr = arr.astype(np.float32 if arr.dtype.itemsize <= 4 else np.float64)
        h = self._ensure_handle()
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(self._device())
        B, n = t.shape
        out = torch.empty((B, self.n_mels, 1 + n // self.hop_length), dtype=t.dtype, device=t.device)
        N.check(N.lib().adm_mel_forward_power(h, N.ptr(t), int(arr.dtype == np.float64), B, n, n, N.ptr(out), N.stream_for(t)))
        return out.cpu().numpy()

    def audio_slice_to_image(self, slice: int, ref: Union[float, Callable] = np.max) -> Image.Image:
        """Convert slice of audio to spectrogram (`mel.py:135-151`)."""
        if ref is not np.max:
            raise NotImplementedError("only ref=np.max (the reference's default and only use) is implemented")
        return Image.fromarray(self.audio_slices_to_images([self.get_audio_slice(slice)])[0])

    def images_to_audios(self, images, init_phase=None, return_magnitude=False):
        """Batched `image_to_audio`: list of PIL images / (B, y_res, x_res) uint8 -> (B, hop*(x_res-1)) float32."""
        if isinstance(images, np.ndarray):
            arr = images
        else:
            arr = np.stack([np.frombuffer(im.tobytes(), dtype="uint8").reshape((im.height, im.width)) for im in images])
        B, n_mels, frames = arr.shape
        assert n_mels == self.n_mels
        h = self._ensure_handle()
        dev = self._device()
        img = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
        n_bins = 1 + self.n_fft // 2
        if init_phase is None:  # librosa: np.random.default_rng().random(S.shape), unseeded
            phase = torch.rand((B, n_bins, frames), dtype=torch.float64, device=dev)
        else:
            phase = torch.as_tensor(np.ascontiguousarray(init_phase), dtype=torch.float64).reshape(B, n_bins, frames).to(dev)
        out = torch.empty((B, self.hop_length * (frames - 1)), dtype=torch.float32, device=dev)
        mag = torch.empty((B, frames, n_bins), dtype=torch.float64, device=dev) if return_magnitude else None
        pg = C.c_float(0.0)
        N.check(N.lib().adm_mel_inverse(h, N.ptr(img), N.ptr(phase.contiguous()), B, frames, N.ptr(out), N.ptr(mag),
                                        C.byref(pg), N.stream_for(img)))
        self.last_nnls_pg = float(pg.value)
        ps, it = C.c_float(0.0), C.c_int(0)
        N.check(N.lib().adm_mel_last_nnls(h, C.byref(ps), C.byref(it)))
        self.last_nnls_pg_start, self.last_nnls_iterations = float(ps.value), int(it.value)
        if self.last_nnls_pg > 1e-5:
            warnings.warn(f"NNLS solution has projected gradient {self.last_nnls_pg:.3g} > pgtol=1e-5 after "
                          f"{self.last_nnls_iterations} iterations (nnls_max_iter={self.nnls_max_iter})")
        audio = out.cpu().numpy()
        if return_magnitude:
            return audio, mag.cpu().numpy().transpose(0, 2, 1)
        return audio

    def image_to_audio(self, image: Image.Image, init_phase=None) -> np.ndarray:
        """Converts spectrogram to audio (`mel.py:153-168`)."""
        return self.images_to_audios([image], init_phase=None if init_phase is None else np.asarray(init_phase)[None])[0]
