"""Timing of conv_wino4_kernel on 128->128 @256x256, B = 32 (and 256->256 @64x64) under the ADM_WINO_ABL role ablations (barrier per chunk:
compare with ADM_WINO_PAIR=0 ADM_WINO_ABL=0). ABL 12 = consumer without its filter loads beside working producers; ABL 13 is NOT an ablation
(correct results): the filter stream as raw buffer loads with an explicitly uniform resource — measured 1.5-4 % slower than the plain
loads (round 4):  for a in 0 13; do ADM_WINO_PAIR=0 ADM_WINO_ABL=$a ADM_LIB=.../libadm_hip_exp.so python tools/wino_abl_probe.py; done"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
from audiodiffusion import _native, ops  # noqa: E402

_native.load(os.environ.get("ADM_LIB") or None)      # the -DADM_EXPERIMENTS build has the ablation instantiations
_native.check(_native.lib().adm_set_option(b"conv_wino", int(os.environ.get("WINO_MODE", "4"))))
dev = torch.device("cuda:0")
for (C, H, Co) in ((128, 256, 128), (256, 64, 256)):
    x = torch.randn(32, C, H, H, device=dev)
    w = torch.randn(Co, C, 3, 3, device=dev) * 0.02
    wp, wu = ops.pack_conv_weight(w), ops.pack_winograd_weight(w)
    b = torch.zeros(Co, device=dev)
    gn = ops.groupnorm_stats(x, torch.ones(C, device=dev), torch.zeros(C, device=dev), 32, 1e-5)
    f = lambda: ops.conv2d(x, wp, b, 3, gn=gn, act=True, wino=wu)  # noqa: E731
    for _ in range(2):
        f()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        f()
    e.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(e) / 5
    tiles = 32 * (H // 8) * (H // 16) * (Co // 64)
    chunks = tiles / 256 * (C // 8)
    print(f"ABL={os.environ.get('ADM_WINO_ABL', '0')} variant {_native.lib().adm_last_conv_variant()} {C}->{Co}@{H}: {ms:.3f} ms  "
          f"= {ms * 1e3 / chunks:.3f} us per chunk and CU ({ms * 1e3 / chunks * 2.3e3:.0f} cycles at 2.3 GHz; 64 MFMAs = 2048)", flush=True)
