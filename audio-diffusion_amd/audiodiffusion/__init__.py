"""MI355X-native drop-in for the `audiodiffusion` package of teticio/audio-diffusion.

Same public names as the reference (`audiodiffusion/__init__.py:12-140`): `AudioDiffusion`,
`AudioDiffusionPipeline`, `Mel`, `VERSION`. Every arithmetic op on the hot path runs in hand-written HIP
kernels for gfx950 (csrc/, C-ABI in include/adm.h); importing this package fails loudly when the native
library has not been built.
"""
from . import _native

# NOTE: the native library is loaded lazily by _native.lib() at the first op; tests may first call
# _native.load(<emulation build>) — the product default is libadm_hip.so and there is no CPU fallback.

VERSION = "1.5.7"

from typing import Iterable, Tuple  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402

from .audio_encoder import AudioEncoder  # noqa: E402,F401
from .mel import Mel  # noqa: E402,F401
from .pipeline_audio_diffusion import AudioDiffusionPipeline  # noqa: E402
from .schedulers import DDIMScheduler, DDPMScheduler  # noqa: E402,F401
from .unet import UNet2DConditionModel, UNet2DModel  # noqa: E402,F401
from .vae import AutoencoderKL  # noqa: E402,F401



def set_option(name: str, value: int) -> None:
    """Process-wide runtime option of the native library (`adm_set_option`, include/adm.h). Not part of the reference's API; the one a user of the
    drop-in may want: `set_option("wino6", 256)` — the single-sample latency setting of the Winograd F(4x4,3x3) kernel (one
    256x256 sample at a time is 18 % faster with it, a batch of 32 10 % slower than with the default), `set_option("wino6", -1)` to go back.
    `AudioDiffusion` (the single-sample facade) selects that rule — and "single_sample" = 1, the split-K rules for layers whose tiles cannot fill the
    chip with one sample — for ITS model by itself (`UNet2DModel.set_option`, per model)."""
    _native.check(_native.lib().adm_set_option(name.encode(), int(value)))


try:  # progress bars are optional plumbing
    from tqdm.auto import tqdm  # noqa: E402
except Exception:  # pragma: no cover
    tqdm = None


class AudioDiffusion:
    """One-sample convenience front end of the pipeline — the reference's `AudioDiffusion` (`audiodiffusion/__init__.py:15-140`):
    same constructor and method signatures, defaults and return shapes (`(PIL image, (sample_rate, audio))`), every call
    forwarded to `AudioDiffusionPipeline.__call__` with `batch_size=1, return_dict=False`.  `model_id` is a local
    directory in the diffusers layout (no hub access); the MI355X is torch's "cuda" device and there is no CPU mode."""

    def __init__(self, model_id: str = "teticio/audio-diffusion-256", cuda: bool = torch.cuda.is_available(),
                 progress_bar: Iterable = tqdm):
        self.model_id = model_id
        self.pipe = AudioDiffusionPipeline.from_pretrained(model_id)
        if cuda:
            self.pipe.to("cuda")
        # Every call of this front end samples ONE spectrogram (`batch_size=1` below, as the reference forces it): its model runs the
        # single-sample layer rules — the planes whose F(4x4) tiles fill the chip with one sample keep that kernel, the levels below take
        # the F(2x2) kernel's smaller tiles ("wino6" = 256), and where even those leave most CUs idle the input channels of a tile are
        # split over several workgroups ("single_sample" = 1). 256x256: 4.6 instead of 8.1 ms per step. Per-MODEL settings
        # (`adm_unet_set_option`): other pipelines in the process — batched sampling, training — keep the default rules.
        if hasattr(self.pipe.unet, "set_option"):
            self.pipe.unet.set_option("wino6", 256)
            self.pipe.unet.set_option("single_sample", 1)
        self.progress_bar = progress_bar if progress_bar is not None else (lambda it: it)

    def _one(self, **call) -> Tuple[Image.Image, Tuple[int, np.ndarray]]:
        images, (sample_rate, audios) = self.pipe(batch_size=1, return_dict=False, **call)
        return images[0], (sample_rate, audios[0])

    def generate_spectrogram_and_audio(self, steps: int = None, generator: torch.Generator = None,
                                       step_generator: torch.Generator = None, eta: float = 0, noise: torch.Tensor = None,
                                       encoding: torch.Tensor = None) -> Tuple[Image.Image, Tuple[int, np.ndarray]]:
        """Unconditional (or `encoding`-conditioned) sample from noise (`__init__.py:35-68`)."""
        return self._one(steps=steps, generator=generator, step_generator=step_generator, eta=eta, noise=noise,
                         encoding=encoding)

    def generate_spectrogram_and_audio_from_audio(self, audio_file: str = None, raw_audio: np.ndarray = None, slice: int = 0,
                                                  start_step: int = 0, steps: int = None, generator: torch.Generator = None,
                                                  mask_start_secs: float = 0, mask_end_secs: float = 0,
                                                  step_generator: torch.Generator = None, eta: float = 0,
                                                  encoding: torch.Tensor = None, noise: torch.Tensor = None
                                                  ) -> Tuple[Image.Image, Tuple[int, np.ndarray]]:
        """Variation / in- and out-painting of one slice of an audio input (`__init__.py:70-122`): `start_step` keeps the
        input noised to that step, `mask_*_secs` pin the start / end of the spectrogram to the input."""
        return self._one(audio_file=audio_file, raw_audio=raw_audio, slice=slice, start_step=start_step, steps=steps,
                         generator=generator, mask_start_secs=mask_start_secs, mask_end_secs=mask_end_secs,
                         step_generator=step_generator, eta=eta, noise=noise, encoding=encoding)

    @staticmethod
    def loop_it(audio: np.ndarray, sample_rate: int, loops: int = 12) -> np.ndarray:
        """Loop audio on bar boundaries (`__init__.py:124-140`): the stretch from the first tracked beat to the last whole
        bar of four beats, tiled `loops` times; None when no full bar was found.  The beat tracker is `beat.beat_track`, a
        host-side restatement of `librosa.beat.beat_track` (post-processing on the CPU, as in the reference)."""
        from .beat import beat_track
        _, beats = beat_track(y=audio, sr=sample_rate, units="samples")
        beats_in_bar = (len(beats) - 1) // 4 * 4
        if beats_in_bar > 0:
            return np.tile(audio[beats[0]:beats[beats_in_bar]], loops)
        return None
