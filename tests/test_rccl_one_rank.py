"""RCCL on ONE MI355X: every collective call the multi-GPU paths make — bench.py's `init_process_group("nccl", device_id=)`,
per-step `all_gather_into_tensor`, barrier, MAX all-reduce; `GradAllReducer`'s asynchronous bucket all-reduces queued from a
ctypes callback INSIDE `adm_unet_forward_backward`; `sample_sharded(gather=True)` — executed in a 1-rank `nccl` group and
compared bit for bit with the group-less result, so that the 8-GPU scaling run cannot fail on plumbing. (A 1-rank group
proves the calls, stream ordering and buffer handling; it measures nothing about xGMI.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd")); sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
from audiodiffusion import AudioDiffusionPipeline, DDIMScheduler, DDPMScheduler, Mel, UNet2DModel
from audiodiffusion import training as T
from audiodiffusion.distributed import sample_sharded
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
CFG = dict(sample_size=(64, 64), in_channels=1, out_channels=1, layers_per_block=2, block_out_channels=(128, 128, 256, 256, 512, 512),
           down_block_types=("DownBlock2D",) * 4 + ("AttnDownBlock2D", "DownBlock2D"),
           up_block_types=("UpBlock2D", "AttnUpBlock2D") + ("UpBlock2D",) * 4)

def sample():
    pipe = AudioDiffusionPipeline(None, UNet2DModel(**CFG).init_random(0), Mel(), DDPMScheduler()).to(dev)
    pipe.set_progress_bar_config(disable=True)
    return sample_sharded(pipe, global_batch=3, steps=4, seed=5, gather=True)[0].cpu()

def train(with_group):
    m = UNet2DModel(**CFG).init_random(1)
    flat, grads = m.enable_training()
    red = T.GradAllReducer(grads, force=with_group)
    if with_group:
        red.attach(m)
    g = torch.Generator().manual_seed(2)
    x, tgt = torch.randn(2, 1, 64, 64, generator=g).to(dev), torch.randn(2, 1, 64, 64, generator=g).to(dev)
    opt, ema = T.AdamW(flat), T.EMAModel(flat)
    for _ in range(2):
        red.begin_step()
        loss = m.train_step(x, torch.tensor([10, 700]), tgt)
        fired = red.overlapped
        red.start(); red.finish()
        opt.step(grads, clip=T.clip_grad_norm_(grads, 1.0), ema=ema, ema_decay=ema.next_decay())
        m.refresh_weights()
    return float(loss), grads.cpu().clone(), flat.cpu().clone(), fired, len(red.bounds)

ref_img = sample()
ref = train(False)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + os.getpid() % 90), RANK="0", WORLD_SIZE="1",
                  HSA_ENABLE_IPC_MODE_LEGACY="0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
img = sample()
got = train(True)
dist.barrier()
dist.destroy_process_group()
assert torch.equal(img, ref_img), "sample_sharded(gather=True) in a 1-rank nccl group differs"
assert got[0] == ref[0] and torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2]), "training step differs with the all-reduce"
assert got[3] == got[4] >= 2, (got[3], got[4])      # every bucket's all-reduce was queued from inside the reverse pass
print("RCCL_ONE_RANK_OK", got[3])
"""


def test_collectives_in_a_one_rank_nccl_group_change_nothing():
    r = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + WORKER], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_bench_py_itself_with_a_one_rank_nccl_group():
    """The driver's bench, full-size model, with ADM_BENCH_FORCE_PG=1: process group, per-step all_gather, barrier-bracketed
    timing, and the training leg's overlapped bucket all-reduce all run over RCCL."""
    env = dict(os.environ, ADM_BENCH_FORCE_PG="1", MASTER_PORT=str(29600 + os.getpid() % 300))
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "1", "--warmup", "1", "--batch-per-gpu", "2",
                        "--ddim-steps", "3", "--no-cpu-baseline", "--no-configs-leg", "--train-steps", "2", "--train-batch-per-gpu", "2"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and "roofline" in d
    tr = d["train"]
    assert "error" not in tr, tr
    assert tr["allreduce_buckets_overlapped"] == tr["allreduce_buckets"] >= 10      # 454.7 MB of gradients in 25 MB buckets
    assert "error" not in d["mel"], d["mel"]
