"""Mel codec timings on the MI355X (device-resident buffers): forward at B = 256 and inverse at B = 32, default config;
ADM_MEL_FAST=0 selects the generic radix-2 kernels, ADM_MEL_OCC the fast kernel's register allocation."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
from audiodiffusion import Mel, _native as N  # noqa: E402

N.load(os.environ.get("ADM_LIB") or None)
dev = torch.device("cuda:0")
mel = Mel()
h = mel._ensure_handle()
n = mel.slice_size
B = 256
audio = (0.3 * torch.randn(B, n, generator=torch.Generator().manual_seed(3))).to(dev)
frames = 1 + n // mel.hop_length
img = torch.empty((B, mel.n_mels, frames), dtype=torch.uint8, device=dev)
spec = torch.empty((B, mel.n_mels, frames), dtype=torch.float32, device=dev)
st = N.stream_for(audio)


def ev(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


tp = ev(lambda: N.check(N.lib().adm_mel_forward_power(h, N.ptr(audio), 0, B, n, n, N.ptr(spec), st)), 10)
tf = ev(lambda: N.check(N.lib().adm_mel_forward(h, N.ptr(audio), 0, B, n, n, N.ptr(img), st)), 10)
tag = f"FAST={os.environ.get('ADM_MEL_FAST', '1')} OCC={os.environ.get('ADM_MEL_OCC', '2')}"
print(f"{tag} forward power B={B}: {tp:.3f} ms; forward to u8: {tf:.3f} ms = {B * 0.59 / tf:.0f} GB/s algorithmic, "
      f"{B / tf * 1e3:.0f} clips/s; checksum {int(img.long().sum())}", flush=True)
Bi = 32
phase = torch.rand((Bi, 1025, frames), dtype=torch.float64, device=dev)
out = torch.empty((Bi, mel.hop_length * (frames - 1)), dtype=torch.float32, device=dev)
images = img[:Bi].contiguous()
ti = ev(lambda: N.check(N.lib().adm_mel_inverse(h, N.ptr(images), N.ptr(phase), Bi, frames, N.ptr(out), None, None, st)), 3)
print(f"{tag} inverse B={Bi}: {ti:.3f} ms = {Bi / ti * 1e3:.0f} clips/s", flush=True)
