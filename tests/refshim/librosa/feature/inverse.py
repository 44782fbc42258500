from oracle import mel as _m

# librosa draws Griffin-Lim's start phases from an unseeded generator; the fixture needs them fixed, so the generating
# script queues one array of angles per expected call here (consumed in call order)
INIT_PHASES = []


def mel_to_audio(M, sr=22050, n_fft=2048, hop_length=512, n_iter=32):
    mag = _m.mel_to_stft(M, sr, n_fft, 2.0)
    return _m.griffinlim(mag, n_iter, hop_length, n_fft, init_phase=INIT_PHASES.pop(0) if INIT_PHASES else None)
