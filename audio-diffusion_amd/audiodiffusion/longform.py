"""Long-form and interpolation procedures of `notebooks/test_model.ipynb`, as library calls on the native pipeline.

The reference has these only as notebook cells that call `AudioDiffusion` once per clip; here each is one function whose
heavy parts are batched on the device:

* `interpolate`    — cells 32-37 / 41-46: DDIM-invert two spectrograms (ONE batched `encode`), spherical interpolation for a
                     whole grid of weights (`ops.slerp_grid`, one launch pair), ONE batched sampling of all of them.
* `outpaint`       — cell 16: extend a clip segment by segment; every new segment is generated with its first
                     `overlap_secs` pinned to the tail of the previous one (`mask_start_secs`): the per-step mask overwrite
                     is part of the captured denoising graph (csrc/k_sched.hip `sched_step_kernel`), not a Python loop.
* `remix_track`    — cell 20: re-generate a whole track in overlapping slices from `start_step`, re-inserting (peak-
                     normalised) the tail of what was generated into the head of the next slice.

Extra keyword hooks (`noise`, `step_noise`, `init_phases`) exist so that the parity tests can feed the oracle pipeline and
this one identical draws; with their defaults the functions behave as the notebook cells do.
"""
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
from PIL import Image

from . import ops


def slerp_grid(x0: torch.Tensor, x1: torch.Tensor, alphas: Sequence[float]) -> torch.Tensor:
    """`AudioDiffusionPipeline.slerp(x0, x1, alpha)` (`pipeline_audio_diffusion.py:244-258`) for every alpha at once:
    (len(alphas),) + x0.shape."""
    return ops.slerp_grid(x0, x1, alphas)


@torch.no_grad()
def interpolate(pipe, image_a: Image.Image, image_b: Image.Image, alphas: Sequence[float], steps: int = None,
                encode_steps: int = 50, generator: torch.Generator = None, audio: bool = True, init_phases=None):
    """Returns (images, (sample_rate, audios)) with one entry per alpha (notebook cells 32-37)."""
    noise = pipe.encode([image_a, image_b], steps=encode_steps)          # both inversions in one batched loop
    grid = slerp_grid(noise[0], noise[1], alphas)                          # (A, C, H, W)
    return pipe(batch_size=len(alphas), noise=grid, steps=steps, generator=generator, return_dict=False, audio=audio,
                init_phase=init_phases)


def _one(pipe, **kw):
    images, (sr, audios) = pipe(batch_size=1, return_dict=False, **kw)
    return images[0], sr, audios[0]


@torch.no_grad()
def outpaint(pipe, raw_audio: np.ndarray, n_segments: int, overlap_secs: float, start_step: int = 0, steps: int = None,
             generator: torch.Generator = None, step_generator: torch.Generator = None, eta: float = 0,
             noise: Optional[List[torch.Tensor]] = None, step_noise=None, init_phases=None
             ) -> Tuple[np.ndarray, List[Image.Image]]:
    """Notebook cell 16. Returns (track, images): `raw_audio` followed by `n_segments` generated continuations."""
    sr = pipe.mel.get_sample_rate()
    ov = int(overlap_secs * sr)
    track, audio, images = raw_audio, raw_audio, []
    for i in range(n_segments):
        image, sr, audio2 = _one(pipe, raw_audio=audio[-ov:], start_step=start_step, steps=steps, generator=generator,
                                 step_generator=step_generator, eta=eta, mask_start_secs=overlap_secs,
                                 noise=None if noise is None else noise[i].clone(),
                                 step_noise=None if step_noise is None else step_noise[i],
                                 init_phase=None if init_phases is None else init_phases[i])
        images.append(image)
        track = np.concatenate([track, audio2[ov:]])
        audio = audio2
    return track, images


@torch.no_grad()
def remix_track(pipe, track_audio: np.ndarray, overlap_secs: float, start_step: int, seed: int = None, steps: int = None,
                eta: float = 0, noise: Optional[torch.Tensor] = None, step_noise=None, init_phases=None
                ) -> Tuple[np.ndarray, List[Image.Image]]:
    """Notebook cell 20. The generator is re-seeded to the same seed before every slice, as the cell does (`noise` overrides
    the draw: the same tensor for every slice). Returns (track, images)."""
    mel = pipe.mel
    sr = mel.get_sample_rate()
    ov = int(overlap_secs * sr)
    slice_size = mel.x_res * mel.hop_length
    stride = slice_size - ov
    generator = torch.Generator(device=pipe.device)
    seed = generator.seed() if seed is None else seed
    track, images, audio2, not_first = np.array([]), [], None, 0
    for sample in range(len(track_audio) // stride):
        generator.manual_seed(seed)
        audio = np.array(track_audio[sample * stride:sample * stride + slice_size])
        if not_first:
            # Normalize and re-insert generated audio
            audio[:ov] = audio2[-ov:] * np.max(audio[:ov]) / np.max(audio2[-ov:])
        image, sr, audio2 = _one(pipe, raw_audio=audio, start_step=start_step, steps=steps, generator=generator, eta=eta,
                                 mask_start_secs=overlap_secs * not_first,
                                 noise=None if noise is None else noise.clone(),
                                 step_noise=step_noise,
                                 init_phase=None if init_phases is None else init_phases[sample])
        images.append(image)
        track = np.concatenate([track, audio2[ov * not_first:]])
        not_first = 1
    return track, images
