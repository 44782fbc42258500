"""Stand-in for `librosa` (see ../README.md): the functions the reference's Mel class calls, bound to oracle/mel.py."""
from oracle.mel import db_to_power, power_to_db  # noqa: F401

from . import beat, feature  # noqa: F401


def load(path, mono=True, sr=None):
    raise NotImplementedError("file decoding is outside the path (SURVEY §8 M1)")
