"""Event timing of the Winograd convolution on two layers of the bench workload (128 -> 128 @256x256 and 256 + 128 -> 128 @128x128, B = 32; GroupNorm +
SiLU on load, residual), with the cycles per 8-channel chunk and CU beside the MFMA cycles they contain. ADM_LIB=<tools/variants/libadm_X.so>,
ADM_WINO6 / ADM_WINO5 select the build / kernel; PROBE_ONE=1: the first layer only."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
from audiodiffusion import _native, ops  # noqa: E402

_native.load(os.environ.get("ADM_LIB") or None)
dev = torch.device("cuda:0")
shapes = ((128, 0, 256, 128), (256, 128, 128, 128)) if not os.environ.get("PROBE_ONE") else ((128, 0, 256, 128),)
for (C1, C2, H, Co) in shapes:
    x = torch.randn(32, C1, H, H, device=dev)
    x2 = torch.randn(32, C2, H, H, device=dev) if C2 else None
    C = C1 + C2
    w = torch.randn(Co, C, 3, 3, device=dev) * 0.02
    wp, wu = ops.pack_conv_weight(w), ops.pack_winograd_weight(w)
    b = torch.zeros(Co, device=dev)
    res = torch.randn(32, Co, H, H, device=dev)
    gn = ops.groupnorm_stats(x, torch.ones(C, device=dev), torch.zeros(C, device=dev), 32, 1e-5, x2=x2)
    f = lambda: ops.conv2d(x, wp, b, 3, x2=x2, gn=gn, act=True, wino=wu, residual=res)  # noqa: E731
    for _ in range(2):
        f()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        f()
    e.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(e) / 5
    var = _native.lib().adm_last_conv_variant()
    if var == 4316:      # F(4x4): 16x16-pixel x 128-cout workgroup tiles, 72 MFMAs (32 cycles each) per wave and chunk of 8 input channels
        tiles, mfma = 32 * (H // 16) * (H // 16) * (Co // 128), 2 * 72 * 32
    else:
        tiles, mfma = 32 * (H // 8) * (H // 16) * (Co // (128 if var == 4315 else 64)), (4096 if var == 4315 else 2048)
    chunks = tiles / 256 * (C // 8)
    print(f"W5={os.environ.get('ADM_WINO5', '1')} variant {var} {C}->{Co}@{H}: {ms:.3f} ms  "
          f"= {ms * 1e3 / chunks:.3f} us per chunk and CU ({ms * 1e3 / chunks * 2.1e3:.0f} cycles at 2.1 GHz; MFMA {mfma} per SIMD)", flush=True)
