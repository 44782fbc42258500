"""A/B of the Winograd kernels on the MI355X: per layer shape of the 256x256 UNet (B = 32, event-timed) and the whole forward,
for each conv_wino mode given on the command line (default "3 4"). WINO_PROF=1 additionally prints the per-role cycle
accounting of one launch of the largest layer (ADM_WINO_PROF in the library)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
from audiodiffusion import _native, ops  # noqa: E402
from audiodiffusion.unet import UNet2DModel  # noqa: E402

_native.load(os.environ.get("ADM_LIB") or None)      # ADM_LIB: another build (e.g. libadm_hip_exp.so for ADM_WINO_PROF)
dev = torch.device("cuda:0")
B = int(os.environ.get("PROBE_B", "32"))
modes = [int(m) for m in sys.argv[1:]] or [3, 4]
CFG256 = dict(sample_size=256, in_channels=1, out_channels=1, layers_per_block=2,
              block_out_channels=(128, 128, 256, 256, 512, 512),
              down_block_types=("DownBlock2D",) * 4 + ("AttnDownBlock2D", "DownBlock2D"),
              up_block_types=("UpBlock2D", "AttnUpBlock2D") + ("UpBlock2D",) * 4)
SHAPES = [  # (C1, C2, H, W, Cout, up)
    (128, 0, 256, 256, 128, 0), (128, 128, 256, 256, 128, 0), (128, 0, 128, 128, 128, 0), (256, 128, 128, 128, 128, 0),
    (128, 0, 64, 64, 256, 0), (256, 0, 64, 64, 256, 0), (256, 256, 64, 64, 256, 0), (256, 0, 32, 32, 256, 0),
    (512, 0, 16, 16, 512, 0), (512, 512, 16, 16, 512, 0), (128, 0, 128, 128, 128, 1), (256, 0, 32, 32, 256, 1),
]


def ev_time(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


ref_out = {}
for mode in modes:
    _native.check(_native.lib().adm_set_option(b"conv_wino", mode))
    tot = 0.0
    for (C1, C2, H, W, Co, up) in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(C1 + 7 * C2 + H)
        x1 = torch.randn(B, C1, H, W, generator=g).to(dev)
        x2 = torch.randn(B, C2, H, W, generator=g).to(dev) if C2 else None
        w = (torch.randn(Co, C1 + C2, 3, 3, generator=g) * 0.02).to(dev)
        wp, wu = ops.pack_conv_weight(w), ops.pack_winograd_weight(w)
        b = torch.randn(Co, generator=g).to(dev)
        gamma, beta = torch.ones(C1 + C2, device=dev), torch.zeros(C1 + C2, device=dev)
        gn = ops.groupnorm_stats(x1, gamma, beta, 32, 1e-5, x2=x2)
        f = lambda: ops.conv2d(x1, wp, b, 3, x2=x2, up=bool(up), gn=gn, act=True, wino=wu)  # noqa: E731
        out = f()
        var = _native.lib().adm_last_conv_variant()
        ms = ev_time(f)
        tot += ms
        fl = 2.0 * out.numel() * (C1 + C2) * 9
        key = (C1, C2, H, W, Co, up)
        err = ""
        if key in ref_out:
            err = f"  max|d vs mode {modes[0]}| {float((out - ref_out[key]).abs().max()):.2e}"
        else:
            ref_out[key] = out.clone()
        print(f"mode {mode} variant {var} {C1}+{C2}@{H}x{W}->{Co} up{up}: {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TF/s algorithmic "
              f"({fl / ms / 1e9 / 2.25 / 157.3:5.3f} of the fp32 MFMA peak executed){err}", flush=True)
        del x1, x2, w, wp, wu, out
    print(f"mode {mode} sum over shapes {tot:.2f} ms", flush=True)
    m = UNet2DModel(**CFG256).init_random(0)
    x = torch.randn(B, 1, 256, 256, device=dev)
    ms = ev_time(lambda: m(x, torch.tensor(500)), iters=3, warm=2)
    print(f"mode {mode} UNet forward B={B}: {ms:.2f} ms", flush=True)
    del m
_native.check(_native.lib().adm_set_option(b"conv_wino", -1))
