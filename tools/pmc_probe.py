"""Small fixed workload for rocprofv3 --pmc passes: 3x big conv, 3x gn_stats (known byte count: calibrates FETCH_SIZE),
1x stride-2 conv, 1x 1x1 conv, at B=32."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
from audiodiffusion import _native, ops  # noqa: E402

_native.load()
dev = torch.device("cuda:0")
B = 32
x = torch.randn(B, 128, 256, 256, device=dev)
w = torch.randn(128, 128, 3, 3, device=dev) * 0.02
wp = ops.pack_conv_weight(w)
b = torch.zeros(128, device=dev)
gamma, beta = torch.ones(128, device=dev), torch.zeros(128, device=dev)
gn = ops.groupnorm_stats(x, gamma, beta, 32, 1e-5)
for _ in range(3):
    out = ops.conv2d(x, wp, b, 3, gn=gn, act=True)
for _ in range(3):
    ops.groupnorm_stats(x, gamma, beta, 32, 1e-5)
ops.conv2d(x, wp, b, 3, stride=2)
w1 = torch.randn(128, 128, 1, 1, device=dev) * 0.05
ops.conv2d(x, ops.pack_conv_weight(w1), b, 1)
torch.cuda.synchronize()
print("bytes per gn_stats read:", x.numel() * 4, "conv in+out bytes:", x.numel() * 8)
