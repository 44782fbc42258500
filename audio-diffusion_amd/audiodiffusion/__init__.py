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
