"""Fixed small workload for rocprofv3 passes on the blocked-image 16-bit kernels (round 4): 3 launches each of the image writer,
the forward convolution, the data gradient and the weight gradient of a 128->128 3x3 layer @256x256, batch 16 (PROBE_C = channels)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
from audiodiffusion import _native, ops  # noqa: E402

_native.load()
dev = torch.device("cuda:0")
B, C = 16, int(os.environ.get("PROBE_C", "128"))
x = torch.randn(B, C, 256, 256, device=dev)
dy = torch.randn(B, 128, 256, 256, device=dev)
w = torch.randn(128, C, 3, 3, device=dev) * 0.02
wb, wbT = ops.pack_bf16_weight(w), ops.pack_bf16_weight(w, transposed=True)
b = torch.zeros(128, device=dev)
gn = ops.groupnorm_stats(x, torch.ones(C, device=dev), torch.zeros(C, device=dev), 32, 1e-5)
for _ in range(3):
    img = ops.blocked_image(x, gn=gn, act=True)
    dimg = ops.blocked_image(dy)
    out = ops.conv2d_bf16_blocked(img, wb, 128, bias=b)
    dx = ops.conv2d_bf16_blocked(dimg, wbT, C)
    dW = ops.conv2d_wgrad_bf16_blocked(img, dimg)
torch.cuda.synchronize()
print("done")
