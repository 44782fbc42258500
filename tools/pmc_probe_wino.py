"""Fixed workload for rocprofv3 --pmc passes on the Winograd conv: 3x 128->128 @256x256 batch 32 with GN+SiLU on load."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
from audiodiffusion import _native, ops  # noqa: E402

_native.load()
mode = int(os.environ.get("WINO_MODE", "3"))
_native.check(_native.lib().adm_set_option(b"conv_wino", mode))
dev = torch.device("cuda:0")
B = 32
x = torch.randn(B, 128, 256, 256, device=dev)
w = torch.randn(128, 128, 3, 3, device=dev) * 0.02
wp = ops.pack_conv_weight(w)
wu = ops.pack_winograd_weight(w)
b = torch.zeros(128, device=dev)
gamma, beta = torch.ones(128, device=dev), torch.zeros(128, device=dev)
gn = ops.groupnorm_stats(x, gamma, beta, 32, 1e-5)
for _ in range(3):
    out = ops.conv2d(x, wp, b, 3, gn=gn, act=True, wino=wu)
torch.cuda.synchronize()
print("variant", _native.lib().adm_last_conv_variant())
