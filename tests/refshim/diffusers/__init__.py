"""Stand-in for `diffusers` (see ../README.md): the names the reference imports, bound to the CPU oracle."""
import torch

from oracle.schedulers import DDIMScheduler, DDPMScheduler  # noqa: F401
from oracle.unet import UNet2DModel  # noqa: F401
from oracle.unet_condition import UNet2DConditionModel  # noqa: F401
from oracle.vae import AutoencoderKL  # noqa: F401

from .configuration_utils import ConfigMixin  # noqa: F401
from .utils import BaseOutput


class ModelMixin(torch.nn.Module):
    pass


class Mel:
    """audio_encoder.py:3,65 takes the class from diffusers (where the reference's own mel.py was upstreamed): hand it the
    reference's class."""

    def __new__(cls, *a, **k):
        from audiodiffusion.mel import Mel as ReferenceMel
        return ReferenceMel(*a, **k)


class AudioPipelineOutput(BaseOutput):
    def __init__(self, audios):
        super().__init__(audios=audios)


class ImagePipelineOutput(BaseOutput):
    def __init__(self, images):
        super().__init__(images=images)


class DiffusionPipeline:
    """register_modules / progress_bar / device: the three members the reference's pipeline uses from its base class."""

    def register_modules(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def progress_bar(self, iterable):
        return iterable

    @property
    def device(self):
        return torch.device("cpu")

    @classmethod
    def from_pretrained(cls, *a, **k):
        raise NotImplementedError("no hub access in the build container")
