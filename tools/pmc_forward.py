"""One eager UNet forward at the bench shape (B=32, 256x256) for rocprofv3 --pmc passes: the dispatches of the SECOND
forward (the first one packs weights / plans the arena) are the per-launch population bench.py's roofline leg averages."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
from audiodiffusion import UNet2DModel, _native  # noqa: E402

_native.load()
CFG = dict(sample_size=(256, 256), in_channels=1, out_channels=1, layers_per_block=2,
           block_out_channels=(128, 128, 256, 256, 512, 512),
           down_block_types=("DownBlock2D",) * 4 + ("AttnDownBlock2D", "DownBlock2D"),
           up_block_types=("UpBlock2D", "AttnUpBlock2D") + ("UpBlock2D",) * 4)
dev = torch.device("cuda:0")
m = UNet2DModel(**CFG).init_random(0)
B = int(os.environ.get("PMC_B", "32"))
x = torch.randn(B, 1, 256, 256, device=dev)
for _ in range(2):
    m(x, torch.tensor(500))
torch.cuda.synchronize()
print("forwards done")
