"""Round 4: event timings of the blocked-image 16-bit training convolutions (k_conv_bf16b.hip) beside round 2's kernels, B = 16.
Per layer shape: image writer (apply pass), forward, data gradient, weight gradient (new: from the blocked images; old: fused
load path), with a cross-check of the two families on the same inputs."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
from audiodiffusion import _native, ops  # noqa: E402

_native.load()
dev = torch.device("cuda:0")
B = int(os.environ.get("PROBE_B", "16"))


def timed(f, reps=5):
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def rel(a, b):
    return float((a - b).double().norm() / b.double().norm())


shapes = [(128, 128, 256), (256, 128, 256), (128, 128, 128), (256, 256, 64), (512, 256, 64), (512, 512, 32)]
if os.environ.get("PROBE_SHAPES"):
    shapes = [tuple(int(v) for v in s.split(",")) for s in os.environ["PROBE_SHAPES"].split(";")]
for (C, Co, HW) in shapes:
    x = torch.randn(B, C, HW, HW, device=dev)
    dy = torch.randn(B, Co, HW, HW, device=dev)
    w = torch.randn(Co, C, 3, 3, device=dev) * (C * 9) ** -0.5
    wp, wb, wbT, wpT = ops.pack_conv_weight(w), ops.pack_bf16_weight(w), ops.pack_bf16_weight(w, transposed=True), ops.pack_conv_weight_T(w)
    gn = ops.groupnorm_stats(x, torch.ones(C, device=dev), torch.zeros(C, device=dev), 32, 1e-5)
    b = torch.zeros(Co, device=dev)
    fl = 2.0 * B * Co * C * 9 * HW * HW
    tf = lambda us: fl / us / 1e6  # noqa: E731
    # --- new family
    t_apply = timed(lambda: ops.blocked_image(x, gn=gn, act=True))
    img = ops.blocked_image(x, gn=gn, act=True)
    t_fwd = timed(lambda: ops.conv2d_bf16_blocked(img, wb, Co, bias=b))
    out_new = ops.conv2d_bf16_blocked(img, wb, Co, bias=b)
    t_dyimg = timed(lambda: ops.blocked_image(dy))
    dimg = ops.blocked_image(dy)
    t_dg = timed(lambda: ops.conv2d_bf16_blocked(dimg, wbT, C))
    dx_new = ops.conv2d_bf16_blocked(dimg, wbT, C)
    t_wg = timed(lambda: ops.conv2d_wgrad_bf16_blocked(img, dimg))
    dW_new = ops.conv2d_wgrad_bf16_blocked(img, dimg)
    # --- round 2's kernels (level 1 dispatch)
    _native.check(_native.lib().adm_set_option(b"conv_bf16", 2))
    try:
        t_fwd_old = timed(lambda: ops.conv2d(x, wp, b, 3, gn=gn, act=True, bf16=wb))
        out_old = ops.conv2d(x, wp, b, 3, gn=gn, act=True, bf16=wb)
        v_old = _native.lib().adm_last_conv_variant()
        t_dg_old = timed(lambda: ops.conv2d(dy, wpT, None, 3, bf16=wbT))
        dx_old = ops.conv2d(dy, wpT, None, 3, bf16=wbT)
        t_wg_old = timed(lambda: ops.conv2d_wgrad(x, dy, Co, 3, gn=gn, act=True))
        dW_old = ops.conv2d_wgrad(x, dy, Co, 3, gn=gn, act=True)
    finally:
        _native.check(_native.lib().adm_set_option(b"conv_bf16", 0))
    print(f"{C:4d}->{Co:4d} @{HW:3d} B={B}: apply {t_apply:7.1f} us | fwd new {t_fwd:7.1f} ({tf(t_fwd):6.0f} TF/s) old {t_fwd_old:7.1f} ({tf(t_fwd_old):6.0f}, variant {v_old}) "
          f"| dy image {t_dyimg:7.1f} | dgrad new {t_dg:7.1f} ({tf(t_dg):6.0f}) old {t_dg_old:7.1f} ({tf(t_dg_old):6.0f}) "
          f"| wgrad new {t_wg:7.1f} ({tf(t_wg):6.0f}) old {t_wg_old:7.1f} ({tf(t_wg_old):6.0f}) "
          f"| new vs old: fwd {rel(out_new, out_old):.1e} dgrad {rel(dx_new, dx_old):.1e} wgrad {rel(dW_new, dW_old):.1e}", flush=True)
