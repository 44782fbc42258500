"""Per-phase cycle accounting (ADM_WGRAD_PROF=1) and event timing of the generic stride-2 fp32 weight-gradient kernel, B = 16."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
from audiodiffusion import _native, ops  # noqa: E402

_native.load()
dev = torch.device("cuda:0")
for (C, Co, HW) in [(128, 128, 256), (256, 256, 64)]:
    x = torch.randn(16, C, HW, HW, device=dev)
    dy = torch.randn(16, Co, HW // 2, HW // 2, device=dev)
    f = lambda: ops.conv2d_wgrad(x, dy, Co, 3, stride=2)  # noqa: E731
    f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 3 * 1e3
    fl = 2.0 * 16 * Co * C * 9 * (HW // 2) ** 2
    print(f"wgrad stride 2 {C}->{Co}@{HW}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s", flush=True)
