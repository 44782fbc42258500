"""Re-capture churn: N single-step sampling calls on a tiny model with NO host synchronisation in between, each on a fresh sample tensor (a new
capture key, so the previous call's executable graph is destroyed while its replay may still be in flight). PROBE_CALLS (default 2000).
PROBE_DESTROYER=k: a second Python thread destroys k other models (planned, with captured loops of their own) one by one WHILE the main thread
is inside the native call (ctypes releases the GIL) — what the cyclic garbage collector does to a test process when another thread wakes it
(tqdm's monitor thread). ADM_GRAPH_DRAIN=0 / ADM_DESTROY_LOCK=0 remove the two guards in unet_exec.hip for the A/B."""
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
from audiodiffusion import AudioDiffusionPipeline, DDIMScheduler, Mel, UNet2DModel  # noqa: E402

cfg = dict(sample_size=16, in_channels=1, out_channels=1, layers_per_block=1, block_out_channels=(32, 64),
           down_block_types=("DownBlock2D", "AttnDownBlock2D"), up_block_types=("AttnUpBlock2D", "UpBlock2D"))
dev = torch.device("cuda:0")


def pipe_of(seed):
    p = AudioDiffusionPipeline(None, UNet2DModel(**cfg).init_random(seed), Mel(x_res=16, y_res=16), DDIMScheduler()).to(dev)
    p.set_progress_bar_config(disable=True)
    p.scheduler.set_timesteps(50)
    return p


pipe = pipe_of(0)
y = torch.randn(2, 1, 16, 16, device=dev)
n = int(os.environ.get("PROBE_CALLS", "2000"))
victims = []
for i in range(int(os.environ.get("PROBE_DESTROYER", "0"))):
    v = pipe_of(i + 1)
    v._denoise(y, 0, 0.0, None, None, 0, 0, stop_step=2)      # planned, captured
    victims.append(v)
torch.cuda.synchronize()
stop = False


def destroyer():
    while victims and not stop:
        v = victims.pop()
        v.unet._free()                                         # adm_unet_destroy from THIS thread
        del v
        time.sleep(0.002)


t = threading.Thread(target=destroyer)
t.start()
for i in range(n):
    y, _ = pipe._denoise(y, i % 50, 0.0, None, None, 0, 0, stop_step=i % 50 + 1)
stop = True
t.join()
torch.cuda.synchronize()
print(f"{n} re-captures without a host synchronisation, {os.environ.get('PROBE_DESTROYER', '0')} models destroyed from a second thread "
      f"meanwhile ({len(victims)} left): ok (finite: {bool(torch.isfinite(y).all())})")
