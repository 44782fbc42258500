"""Per-launch records of one eager UNet forward at the bench workload (B = 32, 256x256): kind, variant, ms, TF/s, GB/s."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
from audiodiffusion import UNet2DModel, _native as N  # noqa: E402
from bench import CFG256  # noqa: E402

N.load()
dev = torch.device("cuda:0")
B = int(os.environ.get("PROBE_B", "32"))
unet = UNet2DModel(**CFG256).init_random(0).to(dev)
x = torch.randn(B, 1, 256, 256, device=dev)
out = torch.empty_like(x)
cap = 1024
recs = (N.OpProfile * cap)()
n = C.c_int(0)
best = None
for _ in range(3):
    N.check(N.lib().adm_unet_profile(unet._ensure_handle(), N.ptr(x), 500.0, N.ptr(out), B, recs, cap, C.byref(n), N.stream_for(x)))
    rows = [(r.kind, r.variant, r.ms, r.flops, r.bytes) for r in recs[: n.value]]
    if best is None or sum(r[2] for r in rows) < sum(r[2] for r in best):
        best = rows
print(f"forward {sum(r[2] for r in best):.3f} ms, {len(best)} launches")
for i, (k, v, ms, fl, by) in enumerate(best):
    print(f"{i:3d} kind {k} var {v:5d} {ms * 1e3:9.1f} us  {fl / ms / 1e9 if ms else 0:7.1f} TF/s  {by / ms / 1e6 if ms else 0:7.1f} GB/s  {fl / 1e9:8.2f} GF")
