"""Stand-in for `librosa` (see ../README.md): the functions the reference's Mel class calls, bound to oracle/mel.py."""
from oracle.mel import db_to_power, power_to_db  # noqa: F401

from . import beat, feature, util  # noqa: F401


def load(path, mono=True, sr=None):
    """16-bit PCM mono WAV at the requested rate: what librosa.load returns for such a file is the samples / 32768 as
    float32 (soundfile's conversion), with no resampling.  Anything else is outside the path."""
    import wave

    import numpy as np
    with wave.open(path, "rb") as w:
        if w.getnchannels() != 1 or w.getsampwidth() != 2 or (sr is not None and w.getframerate() != sr):
            raise NotImplementedError("stand-in decodes mono 16-bit WAV at the target rate only")
        data = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
        return (data.astype(np.float32) / 32768.0), w.getframerate()
