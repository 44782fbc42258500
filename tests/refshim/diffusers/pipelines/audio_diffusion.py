"""diffusers.pipelines.audio_diffusion.Mel is the reference's own mel.py upstreamed: hand the scripts the reference's class."""
from audiodiffusion.mel import Mel  # noqa: F401  (resolves to /root/reference/audiodiffusion/mel.py in the generator process)
