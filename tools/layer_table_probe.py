"""Per-launch table of ONE eager UNet forward at the bench workload (256x256; PROBE_B samples, default 32): every convolution launch with its
variant, duration (best of PROBE_N eager profiles, HIP events around each launch), algorithmic FLOPs / bytes and the rates they give — the
Winograd launches also as EXECUTED fp32-MFMA TF/s (algorithmic / 4 for F(4x4), / 2.25 for F(2x2)) against the 157.3 TF peak.
  ADM_LIB=<other build> PROBE_B=1 python tools/layer_table_probe.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
from audiodiffusion import UNet2DModel, _native as N  # noqa: E402
from bench import CFG256  # noqa: E402

N.load(os.environ.get("ADM_LIB") or None)
dev = torch.device("cuda:0")
B = int(os.environ.get("PROBE_B", "32"))
unet = UNet2DModel(**CFG256).init_random(0)
x = torch.randn(B, 1, 256, 256, generator=torch.Generator().manual_seed(0)).to(dev)
out = torch.empty_like(x)
recs = (N.OpProfile * 1024)()
n = C.c_int(0)
best = None
for _ in range(int(os.environ.get("PROBE_N", "5"))):
    N.check(N.lib().adm_unet_profile(unet._ensure_handle(), N.ptr(x), 500.0, N.ptr(out), B, recs, 1024, C.byref(n), N.stream_for(x)))
    rows = [(r.kind, r.variant, r.ms, r.flops, r.bytes) for r in recs[: n.value]]
    if best is None:
        best = rows
    else:
        best = [b if b[2] <= r[2] else r for b, r in zip(best, rows)]
tot = sum(r[2] for r in best)
print(f"B = {B}: forward {tot:.3f} ms over {len(best)} launches")
by = {}
for i, (kind, var, ms, fl, by_) in enumerate(best):
    by.setdefault(var, [0, 0.0, 0.0])
    by[var][0] += 1; by[var][1] += ms; by[var][2] += fl
    if fl > 0 and ms > 0.02:
        div = 4.0 if var == 4316 else (2.25 if var // 100 == 43 else 1.0)
        print(f"  #{i:3d} kind {kind:2d} variant {var:5d}  {ms * 1e3:8.1f} us  {fl / 1e9:8.2f} GFLOP  {by_ / 1e6:8.1f} MB  "
              f"{fl / ms / 1e9:7.1f} TF/s alg  {fl / div / ms / 1e9 / 157.3:5.3f} of fp32 MFMA peak executed  {by_ / ms / 1e9:6.2f} TB/s")
for var, (cnt, ms, fl) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"  variant {var:5d}: {cnt:3d} launches {ms:8.3f} ms" + (f"  {fl / ms / 1e9:7.1f} TF/s alg" if fl > 0 and ms > 0 else ""))
