import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
import numpy as np, torch
import test_reference_pin as T
from native_backend import select
dev = select("hip")
from audiodiffusion import AudioDiffusionPipeline, DDIMScheduler, Mel, UNet2DModel
from audiodiffusion.longform import outpaint
unet = UNet2DModel(**dict(T.UNET_CFG, sample_size=T.NB_HW)).load_state_dict(T.UNET_SD)
p = AudioDiffusionPipeline(None, unet, Mel(**T.MEL_NB), DDIMScheduler()); p.set_progress_bar_config(disable=True)
ph = T.notebook_phases().astype(np.float64)
track, images = outpaint(p, T.notebook_clip(1.0, 3), 12, T.NB_OVERLAP, start_step=0, noise=[x.to(dev) for x in T._outpaint_noise(12)], init_phases=ph[:12, None])
track = np.asarray(track, dtype=np.float32)
want = T.Z["N:outpaint_track_every4"][: (len(track) + 3) // 4]
d = np.abs(track[::4] - want)
per = T.MEL_NB["hop_length"] * (T.MEL_NB["x_res"] - 1) - T.NB_OVERLAP * T.NB_SR
print("outpaint: max rel dev", d.max() / np.abs(want).max(), "first deviating sample*4", int(np.argmax(d > 1e-4 * np.abs(want).max())) * 4, "samples per segment", per)
a, b = np.asarray(images[-1]).astype(int), T.Z["N:outpaint_last_image"].astype(int)
print("last image: max diff", np.abs(a - b).max(), "identical", (a == b).mean())
