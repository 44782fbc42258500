from oracle import mel as _m

from . import inverse  # noqa: F401


def melspectrogram(y=None, sr=22050, n_fft=2048, hop_length=512, n_mels=128):
    return _m.melspectrogram(y, sr, n_fft, hop_length, n_mels)
