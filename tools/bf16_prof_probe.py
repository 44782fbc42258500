"""Per-phase cycle accounting (ADM_BF16_PROF=1) and event timings of the bf16 forward and weight-gradient kernels, B = 16."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
from audiodiffusion import _native, ops  # noqa: E402

_native.load()
_native.check(_native.lib().adm_set_option(b"conv_bf16", 2))
dev = torch.device("cuda:0")


def timed(f, reps=5):
    f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (C, Co, HW) in [(128, 128, 256), (256, 128, 256), (256, 256, 64), (512, 512, 16)]:
    x = torch.randn(16, C, HW, HW, device=dev)
    dy = torch.randn(16, Co, HW, HW, device=dev)
    w = torch.randn(Co, C, 3, 3, device=dev) * 0.02
    wp, wb = ops.pack_conv_weight(w), ops.pack_bf16_weight(w)
    gn = ops.groupnorm_stats(x, torch.ones(C, device=dev), torch.zeros(C, device=dev), 32, 1e-5)
    b = torch.zeros(Co, device=dev)
    fl = 2.0 * 16 * Co * C * 9 * HW * HW
    us = timed(lambda: ops.conv2d(x, wp, b, 3, gn=gn, act=True, bf16=wb))
    print(f"fwd   {C}->{Co}@{HW}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s", flush=True)
    us = timed(lambda: ops.conv2d_wgrad(x, dy, Co, 3, gn=gn, act=True))
    print(f"wgrad {C}->{Co}@{HW}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s (incl. split-K reduction + workspace allocation)", flush=True)
