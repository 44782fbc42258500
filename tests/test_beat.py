"""`AudioDiffusion.loop_it` (audiodiffusion/__init__.py:124-140) and its beat tracker on synthetic click tracks: the tempo
is recovered, beats land on the clicks, the loop is a whole number of bars."""
import numpy as np
import pytest

from native_backend import select


def _clicks(bpm, secs, sr=22050, seed=0):
    rng = np.random.default_rng(seed)
    y = 0.005 * rng.standard_normal(int(secs * sr)).astype(np.float32)
    period = 60.0 / bpm
    t = np.arange(int(0.03 * sr)) / sr
    click = (np.exp(-t * 150) * np.sin(2 * np.pi * 1500 * t)).astype(np.float32)
    times = np.arange(0.25, secs - 0.1, period)
    for s in (times * sr).astype(int):
        y[s:s + len(click)] += click[: len(y) - s]
    return y, times


@pytest.mark.parametrize("bpm", [90.0, 120.0, 150.0])
def test_beats_follow_a_click_track(bpm):
    select("emu")
    from audiodiffusion.beat import beat_track
    sr = 22050
    y, times = _clicks(bpm, 12.0, sr)
    est, beats = beat_track(y, sr, units="time")
    assert abs(est - bpm) / bpm < 0.04, est
    assert len(beats) >= 0.7 * len(times)
    d = np.abs(beats[:, None] - times[None, :]).min(axis=1)
    assert np.median(d) < 0.05                      # within two hops (46 ms) of a click
    assert np.allclose(np.diff(beats), 60.0 / bpm, atol=0.06)


def test_loop_it_returns_whole_bars_or_none():
    select("emu")
    from audiodiffusion import AudioDiffusion
    from audiodiffusion.beat import beat_track
    sr = 22050
    y, _ = _clicks(120.0, 10.0, sr)
    out = AudioDiffusion.loop_it(y, sr, loops=3)
    _, beats = beat_track(y, sr, units="samples")
    bars = (len(beats) - 1) // 4 * 4
    assert bars >= 4 and out is not None
    assert len(out) == 3 * (beats[bars] - beats[0])
    assert np.array_equal(out[: beats[bars] - beats[0]], y[beats[0]:beats[bars]])
    assert AudioDiffusion.loop_it(np.zeros(sr, dtype=np.float32), sr) is None           # silence: no beats
