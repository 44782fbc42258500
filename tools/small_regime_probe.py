"""Per-launch records of one eager UNet forward in the latency-bound regimes (BASELINE configs 1, 2, 4 and B = 1):
PROBE = "res,B[;res,B...]" (default "32,16;64,1;256,1"). Prints the ops sorted by time and the captured-loop step time next to it."""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
from audiodiffusion import AudioDiffusionPipeline, DDIMScheduler, Mel, UNet2DModel, _native as N  # noqa: E402
from bench import CFG256  # noqa: E402

N.load()
dev = torch.device("cuda:0")
KIND = {0: "gn_stats", 1: "conv", 2: "attention", 3: "conv_small", 4: "temb"}
for spec in os.environ.get("PROBE", "32,16;64,1;256,1").split(";"):
    res, B = (int(v) for v in spec.split(","))
    cfg = dict(CFG256, sample_size=res)
    unet = UNet2DModel(**cfg).init_random(0)
    if os.environ.get("PROBE_RULE"):             # the model's own F(4x4) layer rule (UNet2DModel.set_option; AudioDiffusion sets 256)
        unet.set_option("wino6", int(os.environ["PROBE_RULE"]))
    if os.environ.get("PROBE_KSPLIT"):           # the model's own split-K rule of the 64-cout F(2x2) kernel (AudioDiffusion sets 1)
        unet.set_option("single_sample", int(os.environ["PROBE_KSPLIT"]))
    x = torch.randn(B, 1, res, res, device=dev)
    out = torch.empty_like(x)
    cap = 1024
    recs = (N.OpProfile * cap)()
    n = C.c_int(0)
    best = None
    for _ in range(4):
        N.check(N.lib().adm_unet_profile(unet._ensure_handle(), N.ptr(x), 500.0, N.ptr(out), B, recs, cap, C.byref(n), N.stream_for(x)))
        rows = [(i, r.kind, r.variant, r.ms, r.flops, r.bytes) for i, r in enumerate(recs[: n.value])]
        if best is None or sum(r[3] for r in rows) < sum(r[3] for r in best):
            best = rows
    tot = sum(r[3] for r in best)
    pipe = AudioDiffusionPipeline(None, unet, Mel(), DDIMScheduler()).to(dev)
    pipe.set_progress_bar_config(disable=True)
    pipe.scheduler.set_timesteps(50)
    pipe._denoise(x, 0, 0.0, None, None, 0, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe._denoise(x, 0, 0.0, None, None, 0, 0)
    torch.cuda.synchronize()
    loop = (time.perf_counter() - t0) / 50
    print(f"== {res}x{res} B={B}: eager sum of launches {tot:.3f} ms ({len(best)} launches), captured loop {loop * 1e3:.3f} ms/step")
    by = {}
    for i, k, v, ms, fl, b in best:
        e = by.setdefault((k, v), [0, 0.0, 0.0])
        e[0] += 1; e[1] += ms; e[2] += fl
    for (k, v), e in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print(f"   {KIND.get(k, k):10s} var {v:5d}: {e[0]:3d} launches {e[1] * 1e3:9.1f} us  ({e[1] / tot * 100:5.1f} %)  {e[2] / 1e9:9.2f} GF")
    for i, k, v, ms, fl, b in sorted(best, key=lambda r: -r[3])[:int(os.environ.get("PROBE_TOP", "24"))]:
        print(f"   op {i:3d} {KIND.get(k, k):10s} var {v:5d} {ms * 1e3:8.1f} us {fl / 1e9:9.3f} GF {b / 1e6:9.3f} MB")
    del pipe, unet
