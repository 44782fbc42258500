"""One number for A/B runs on ONE box (box-to-box clock spread is +-2 %): UNet forward at the bench workload (B = 32, 256x256), best of
N eager profiles — total of all launches and of the dominant (Winograd) kernel. Environment switches of the library select the
sides:  for v in 0 1 0 1; do ADM_WINO_RES_SPEC=$v python tools/forward_probe.py; done"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
from audiodiffusion import UNet2DModel, _native as N  # noqa: E402
from bench import CFG256  # noqa: E402

N.load(os.environ.get("ADM_LIB") or None)      # ADM_LIB=<path to another build of libadm_hip.so>: A/B of two builds on one box
dev = torch.device("cuda:0")
B = int(os.environ.get("PROBE_B", "32"))
unet = UNet2DModel(**CFG256).init_random(0)
x = torch.randn(B, 1, 256, 256, generator=torch.Generator().manual_seed(0)).to(dev)
out = torch.empty_like(x)
recs = (N.OpProfile * 1024)()
n = C.c_int(0)
best = None
for _ in range(int(os.environ.get("PROBE_N", "6"))):
    N.check(N.lib().adm_unet_profile(unet._ensure_handle(), N.ptr(x), 500.0, N.ptr(out), B, recs, 1024, C.byref(n), N.stream_for(x)))
    rows = [(r.kind, r.variant, r.ms) for r in recs[: n.value]]
    tot = sum(r[2] for r in rows)
    if best is None or tot < best[0]:
        best = (tot, sum(r[2] for r in rows if r[1] // 100 == 43), sum(1 for r in rows if r[1] // 100 == 43))
print(f"forward {best[0]:.3f} ms  winograd {best[1]:.3f} ms ({best[2]} launches)  env " +
      " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("ADM_")))
if os.environ.get("PROBE_SAVE"):
    torch.save(out.cpu(), os.environ["PROBE_SAVE"])
