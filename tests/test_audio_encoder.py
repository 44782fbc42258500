"""AudioEncoder (audiodiffusion/audio_encoder.py:62-107): native forward vs the oracle restatement, and `encode` on WAV
files (batched Mel + encoder) vs the reference's per-slice procedure run on the oracle."""
import os
import wave

import numpy as np
import pytest
import torch

from native_backend import BACKENDS, select
from oracle import audio_encoder as oenc


def _write_wav(path, x, sr):
    with wave.open(path, "wb") as w:
        w.setnchannels(1), w.setsampwidth(2), w.setframerate(sr)
        w.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("n,hw", [(3, (16, 24)), (9, (8, 40))])
def test_forward_matches_oracle(backend, n, hw):
    dev = select(backend)
    from audiodiffusion.audio_encoder import AudioEncoder
    sd = oenc.random_state_dict(1, y_res=hw[0], x_res=hw[1])
    x = torch.rand((n, 1) + hw, generator=torch.Generator().manual_seed(2))
    want = oenc.forward(sd, x)
    got = AudioEncoder().load_state_dict(sd)(x.to(dev)).cpu()
    assert got.shape == want.shape == (n, 100)
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())


def test_reference_geometry():
    """96 x 216 mel images -> 128 x 12 x 27 = 41472 dense inputs (audio_encoder.py:75), 216 frames per slice."""
    sd = oenc.random_state_dict(0)
    assert sd["dense_block.dense.weight"].shape == (1024, 41472)
    want = sum(ci * 9 + ci * co + co + 2 * co for ci, co in ((1, 32), (32, 64), (64, 128)))       # dw + pw + bias + BN affine
    want += 41472 * 1024 + 1024 + 2 * 1024 + 1024 * 100 + 100
    assert sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k) == want == 42584717


@pytest.mark.parametrize("backend", BACKENDS)
def test_encode_files_and_roundtrip(backend, tmp_path):
    dev = select(backend)
    from audiodiffusion.audio_encoder import AudioEncoder
    from audiodiffusion import Mel
    enc = AudioEncoder()
    # a small mel geometry keeps the emulator run short; the encoder derives everything from the image size
    enc.mel = Mel(x_res=24, y_res=16, sample_rate=4000, n_fft=256, hop_length=64, top_db=80)
    sd = oenc.random_state_dict(3, y_res=16, x_res=24)
    enc.load_state_dict(sd)
    rng = np.random.default_rng(0)
    files = []
    for i, secs in enumerate((1.2, 0.9)):
        t = np.arange(int(4000 * secs)) / 4000
        x = 0.4 * np.sin(2 * np.pi * (300 + 200 * i) * t) + 0.1 * rng.standard_normal(t.size)
        files.append(str(tmp_path / f"a{i}.wav"))
        _write_wav(files[-1], x, 4000)
    got = enc.encode(files).cpu()
    assert got.shape == (2, 100)
    # the reference's procedure (audio_encoder.py:89-104): one image per slice, /255, forward, mean over the slices
    want = []
    for f in files:
        enc.mel.load_audio(f)
        imgs = [np.frombuffer(enc.mel.audio_slice_to_image(s).tobytes(), dtype="uint8").reshape(16, 24) / 255
                for s in range(enc.mel.get_number_of_slices())]
        assert len(imgs) >= 2
        want.append(oenc.forward(sd, torch.Tensor(np.array(imgs))[:, None]).mean(dim=0))
    want = torch.stack(want)
    assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())
    assert enc.encode(files, pool="max").shape == (2, 100)
    enc.save_pretrained(str(tmp_path / "enc"))
    again = AudioEncoder.from_pretrained(str(tmp_path / "enc"))
    again.mel = enc.mel
    assert torch.equal(again.encode(files).cpu(), got)
    assert os.path.exists(tmp_path / "enc" / "config.json")
