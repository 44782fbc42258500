"""Fixed small workload for rocprofv3 passes on the bf16 kernels: 3 launches each of the forward convolution (GN+SiLU on
load), the data gradient and the weight gradient of a 128->128 3x3 layer @256x256, batch 16 (PROBE_C = channels)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
from audiodiffusion import _native, ops  # noqa: E402

_native.load()
_native.check(_native.lib().adm_set_option(b"conv_bf16", 1))
dev = torch.device("cuda:0")
B, C = 16, int(os.environ.get("PROBE_C", "128"))
x = torch.randn(B, C, 256, 256, device=dev)
dy = torch.randn(B, 128, 256, 256, device=dev)
w = torch.randn(128, C, 3, 3, device=dev) * 0.02
wp, wpT = ops.pack_conv_weight(w), ops.pack_conv_weight_T(w)
wb, wbT = ops.pack_bf16_weight(w), ops.pack_bf16_weight(w, transposed=True)
b = torch.zeros(128, device=dev)
gn = ops.groupnorm_stats(x, torch.ones(C, device=dev), torch.zeros(C, device=dev), 32, 1e-5)
for _ in range(3):
    out = ops.conv2d(x, wp, b, 3, gn=gn, act=True, bf16=wb)
    v1 = _native.lib().adm_last_conv_variant()
    dx = ops.conv2d(dy, wpT, None, 3, bf16=wbT)
    dW = ops.conv2d_wgrad(x, dy, 128, 3, gn=gn, act=True)
torch.cuda.synchronize()
print("variant", v1, _native.lib().adm_last_conv_variant())
