"""Timing / parity probe of the conditional UNet (scripts/train_unet.py:139-159 config) at latent resolution on the MI355X:
PROBE_HW (default 64: the 512-resolution latent model), PROBE_B (default 8), PROBE_CHECK=1 compares with the oracle at B = 1.
Not run in round 1 (written after the GPU budget was spent)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "audio-diffusion_amd"), ROOT]
from audiodiffusion import UNet2DConditionModel  # noqa: E402

HW, B = int(os.environ.get("PROBE_HW", "64")), int(os.environ.get("PROBE_B", "8"))
CFG = dict(sample_size=(HW, HW), in_channels=1, out_channels=1, layers_per_block=2, block_out_channels=(128, 256, 512, 512),
           down_block_types=("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",),
           up_block_types=("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3, cross_attention_dim=100, attention_head_dim=8)
dev = torch.device("cuda:0")
m = UNet2DConditionModel(**CFG).init_random(0)
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 1, HW, HW, generator=g).to(dev)
enc = torch.randn(B, 1, 100, generator=g).to(dev)
t = torch.tensor(500)
m(x, t, enc)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    out = m(x, t, enc)["sample"]
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"conditional UNet {HW}x{HW} B={B}: {dt * 1e3:.2f} ms / forward ({m.num_parameters() / 1e6:.1f} M parameters)", flush=True)
if os.environ.get("PROBE_CHECK", "1") == "1":
    from oracle.unet_condition import UNet2DConditionModel as Oracle
    ref = Oracle(**CFG).eval()
    ref.load_state_dict(m.state_dict())
    with torch.no_grad():
        want = ref(x[:1].cpu(), t, enc[:1].cpu())["sample"]
    got = m(x[:1].contiguous(), t, enc[:1].contiguous())["sample"].cpu()
    print(f"parity vs oracle (B=1): max|d| = {float((got - want).abs().max()):.3e} (max|ref| {float(want.abs().max()):.3f})", flush=True)
