"""Fixed workload for rocprofv3 --pmc passes on ONE convolution class of the sampling path ("what bounds the
1x1 / stride-2 / 8x8 class"). PROBE = 1x1 (256 -> 128 @256x256, the up-block conv_shortcut over a virtual concat), s2 (128 -> 128
stride 2 @256x256, Downsample2D) or 8x8 (512 -> 512 3x3 @8x8, split-K). Batch 32, three launches."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
from audiodiffusion import _native, ops  # noqa: E402

_native.load()
dev = torch.device("cuda:0")
B = 32
probe = os.environ.get("PROBE", "1x1")
if probe == "1x1":
    x1, x2 = torch.randn(B, 128, 256, 256, device=dev), torch.randn(B, 128, 256, 256, device=dev)
    w = torch.randn(128, 256, 1, 1, device=dev) * 0.05
    wp, b = ops.pack_conv_weight(w), torch.zeros(128, device=dev)
    run = lambda: ops.conv2d(x1, wp, b, 1, x2=x2)   # noqa: E731
elif probe == "s2":
    x1 = torch.randn(B, 128, 256, 256, device=dev)
    w = torch.randn(128, 128, 3, 3, device=dev) * 0.02
    wp, b = ops.pack_conv_weight(w), torch.zeros(128, device=dev)
    run = lambda: ops.conv2d(x1, wp, b, 3, stride=2)   # noqa: E731
else:
    x1 = torch.randn(B, 512, 8, 8, device=dev)
    w = torch.randn(512, 512, 3, 3, device=dev) * 0.02
    wp, b = ops.pack_conv_weight(w), torch.zeros(512, device=dev)
    run = lambda: ops.conv2d(x1, wp, b, 3)   # noqa: E731
for _ in range(3):
    out = run()
torch.cuda.synchronize()
print("probe", probe, "variant", _native.lib().adm_last_conv_variant())
