"""Per-role cycle accounting of conv_wino4_kernel (one barrier per two chunks) on the headline layer shapes: needs the -DADM_EXPERIMENTS
build (bash audio-diffusion_amd/csrc/build.sh hip exp) and ADM_WINO_PROF=1:
    ADM_WINO_PROF=1 ADM_LIB=audio-diffusion_amd/audiodiffusion/libadm_hip_exp.so python tools/wino_prof_probe.py
The library prints, per launch: [wino3 prof] per-block cycles: consumer total / drain / barrier / epilogue | producer total / drain / barrier / C / B / A."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
from audiodiffusion import _native, ops  # noqa: E402

_native.load(os.environ.get("ADM_LIB") or None)
dev = torch.device("cuda:0")
B = int(os.environ.get("PROBE_B", "32"))
for (C1, H, W, Co) in [(128, 256, 256, 128), (256, 64, 64, 256), (512, 16, 16, 512)]:
    g = torch.Generator(device="cpu").manual_seed(C1 + H)
    x1 = torch.randn(B, C1, H, W, generator=g).to(dev)
    w = (torch.randn(Co, C1, 3, 3, generator=g) * 0.02).to(dev)
    wp, wu = ops.pack_conv_weight(w), ops.pack_winograd_weight(w)
    b = torch.randn(Co, generator=g).to(dev)
    gn = ops.groupnorm_stats(x1, torch.ones(C1, device=dev), torch.zeros(C1, device=dev), 32, 1e-5)
    print(f"--- {C1}->{Co} @{H}x{W}, B = {B}: {C1 // 8} chunks per tile (64 MFMAs = 2048 cycles per chunk and consumer wave)", flush=True)
    for _ in range(3):
        ops.conv2d(x1, wp, b, 3, gn=gn, act=True, wino=wu)
        torch.cuda.synchronize()
    sys.stderr.flush()
