"""Beat tracking for `AudioDiffusion.loop_it` (`audiodiffusion/__init__.py:124-140` calls `librosa.beat.beat_track(y, sr,
units="samples")`).  Host-side numpy, as in the reference: loop_it is post-processing of finished audio on the CPU, not
part of the MI355X hot path.

librosa is not installed here, so this restates the published algorithm of librosa 0.10.2 [3P-recall]:
  onset strength  = median over 128 mel bands of the positive first difference of the dB mel spectrogram
                    (n_fft 2048, hop 512, power_to_db with top_db 80), shifted by lag + n_fft / (2 hop) frames;
  tempo           = arg max over lags of log1p(1e6 * mean autocorrelation tempogram (8 s Hann windows, max-normalised))
                    + a log-normal prior around 120 BPM (sigma 1 octave), lags faster than 320 BPM excluded;
  beats           = Ellis' dynamic programme: onset envelope / std smoothed by a Gaussian of the beat period, transition
                    cost -tightness * log(-lag / period)^2 over lags in [-2 period, -period / 2], backtracked from the
                    last local maximum of the cumulative score above half the median peak, weak leading / trailing beats
                    trimmed.
Unpinned against librosa itself (no fixtures in the reference); `tests/test_beat.py` pins it on synthetic click tracks.
"""
import numpy as np
import scipy.signal

from .mel import slaney_filter_taps, taps_to_dense

HOP, N_FFT, N_MELS = 512, 2048, 128


def _power_to_db(S, amin=1e-10, top_db=80.0):
    log_spec = 10.0 * np.log10(np.maximum(amin, S)) - 10.0 * np.log10(np.maximum(amin, 1.0))
    return np.maximum(log_spec, log_spec.max() - top_db)


def onset_strength(y, sr, hop_length=HOP, n_fft=N_FFT):
    y = np.asarray(y, dtype=np.float32)
    pad = n_fft // 2
    yp = np.pad(y, (pad, pad), mode="constant")
    n_frames = 1 + (len(yp) - n_fft) // hop_length
    idx = np.arange(n_fft)[None, :] + hop_length * np.arange(n_frames)[:, None]
    window = scipy.signal.get_window("hann", n_fft, fftbins=True).astype(np.float32)
    power = np.abs(np.fft.rfft(yp[idx] * window, axis=1)).astype(np.float32) ** 2            # (frames, bins)
    start, count, w32, _ = slaney_filter_taps(sr, n_fft, N_MELS)
    mel = power @ taps_to_dense(start, count, w32, 1 + n_fft // 2).T                            # (frames, mels)
    S = _power_to_db(mel.T)                                                                      # (mels, frames)
    flux = np.maximum(0.0, S[:, 1:] - S[:, :-1])
    env = np.median(flux, axis=0)
    env = np.pad(env, (1 + n_fft // (2 * hop_length), 0), mode="constant")
    return env[: S.shape[1]]


def _autocorrelate_columns(x):
    n = x.shape[0]
    spec = np.fft.fft(x, n=2 * n - 1, axis=0)
    return np.real(np.fft.ifft(np.abs(spec) ** 2, axis=0))[:n]


def tempo(onset_envelope, sr, hop_length=HOP, start_bpm=120.0, std_bpm=1.0, ac_size=8.0, max_tempo=320.0):
    win = int(np.floor(ac_size * sr / hop_length))
    n = len(onset_envelope)
    padded = np.pad(onset_envelope, (win // 2, win // 2), mode="linear_ramp", end_values=(0, 0))
    frames = np.lib.stride_tricks.sliding_window_view(padded, win).T[:, :n]                    # (win, n)
    ac = _autocorrelate_columns(frames * scipy.signal.get_window("hann", win, fftbins=True)[:, None])
    peak = np.abs(ac).max(axis=0, keepdims=True)
    peak[peak < np.finfo(ac.dtype).tiny] = 1.0
    tg = (ac / peak).mean(axis=1)
    bpms = np.full(win, np.inf)
    bpms[1:] = 60.0 * sr / (hop_length * np.arange(1.0, win))
    with np.errstate(divide="ignore"):
        logprior = -0.5 * ((np.log2(bpms) - np.log2(start_bpm)) / std_bpm) ** 2
    logprior[: int(np.argmax(bpms < max_tempo))] = -np.inf
    return float(bpms[int(np.argmax(np.log1p(1e6 * tg) + logprior))])


def _local_max(x):
    xp = np.pad(x, 1, mode="edge")
    return (x > xp[:-2]) & (x >= xp[2:])


def track_beats(onset_envelope, bpm, frame_rate, tightness=100.0, trim=True):
    period = int(round(60.0 * frame_rate / bpm))
    norm = onset_envelope.std(ddof=1)
    onsets = onset_envelope / norm if norm > 0 else onset_envelope
    gauss = np.exp(-0.5 * (np.arange(-period, period + 1) * 32.0 / period) ** 2)
    localscore = scipy.signal.convolve(onsets, gauss, "same")
    # dynamic programme
    lags = np.arange(-2 * period, -int(round(period / 2)) + 1, dtype=int)
    cost = -tightness * np.log(-lags / period) ** 2
    backlink = np.zeros(len(localscore), dtype=int)
    cumscore = np.zeros(len(localscore))
    threshold = 0.01 * localscore.max()
    first = True
    for i, score in enumerate(localscore):
        prev = i + lags
        skip = int(np.clip(-prev[0], 0, len(lags)))               # candidates that would reach before frame 0
        cand = cost.copy()
        cand[skip:] += cumscore[prev[skip:]]
        best = int(np.argmax(cand))
        cumscore[i] = score + cand[best]
        if first and score < threshold:
            backlink[i] = -1
        else:
            backlink[i] = prev[best]
            first = False
    peaks = _local_max(cumscore)
    med = np.median(cumscore[peaks])
    beats = [int(np.argwhere(cumscore * peaks * 2 > med).max())]
    while backlink[beats[-1]] >= 0:
        beats.append(int(backlink[beats[-1]]))
    beats = np.array(beats[::-1], dtype=int)
    smooth = scipy.signal.convolve(localscore[beats], scipy.signal.get_window("hann", 5, fftbins=False), "same")
    thr = 0.5 * np.sqrt((smooth ** 2).mean()) if trim else 0.0
    valid = np.argwhere(smooth > thr)
    return beats[int(valid.min()): int(valid.max())]


def beat_track(y, sr, units="samples", hop_length=HOP, start_bpm=120.0, tightness=100.0):
    """-> (tempo in BPM, beat positions in `units` = "frames" | "samples" | "time")."""
    env = onset_strength(y, sr, hop_length)
    if not env.any():
        return 0.0, np.array([], dtype=int)
    bpm = tempo(env, sr, hop_length, start_bpm)
    beats = track_beats(env, bpm, float(sr) / hop_length, tightness)
    if units == "samples":
        beats = beats * hop_length
    elif units == "time":
        beats = beats * hop_length / float(sr)
    elif units != "frames":
        raise ValueError(f"Invalid unit type: {units}")
    return bpm, beats
