"""Event-timed comparison on a few UNet layer shapes (batch 16): the fused-load bf16 conv (GroupNorm + SiLU + rounding in the
patch load path) against the blocked-activation prototype (one GroupNorm-apply pass writing xb[n][C/8][H][W][8] bf16, then a
convolution with no conversion work).  Not run in round 1."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
from audiodiffusion import _native, ops  # noqa: E402

_native.load()
dev = torch.device("cuda:0")


def timed(fn, n=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (C, Co, HW) in [(128, 128, 256), (256, 128, 256), (256, 256, 64), (512, 512, 16)]:
    x = torch.randn(16, C, HW, HW, device=dev)
    w = torch.randn(Co, C, 3, 3, device=dev) * 0.02
    wp, wb = ops.pack_conv_weight(w), ops.pack_bf16_weight(w)
    gn = ops.groupnorm_stats(x, torch.ones(C, device=dev), torch.zeros(C, device=dev), 32, 1e-5)
    b = torch.zeros(Co, device=dev)
    _native.check(_native.lib().adm_set_option(b"conv_bf16", 1))
    fused = timed(lambda: ops.conv2d(x, wp, b, 3, gn=gn, act=True, bf16=wb))
    ref = ops.conv2d(x, wp, b, 3, gn=gn, act=True, bf16=wb)
    _native.check(_native.lib().adm_set_option(b"conv_bf16", 0))
    apply_us = timed(lambda: ops.gn_apply_bf16_blocked(x, gn=gn, act=True))
    xb = ops.gn_apply_bf16_blocked(x, gn=gn, act=True)
    conv_us = timed(lambda: ops.conv2d_bf16_blocked(xb, wb, b, Co))
    out = ops.conv2d_bf16_blocked(xb, wb, b, Co)
    fl = 2.0 * 16 * Co * C * 9 * HW * HW
    print(f"{C}->{Co}@{HW}: fused-load {fused:8.1f} us ({fl / fused / 1e6:6.1f} TF/s) | apply pass {apply_us:7.1f} us + blocked conv "
          f"{conv_us:8.1f} us ({fl / conv_us / 1e6:6.1f} TF/s) | identical: {bool(torch.equal(out, ref))}", flush=True)
