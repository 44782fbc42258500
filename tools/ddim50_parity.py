"""DDIM-50 at 256x256 end to end, product (MI355X) against the CPU oracle — measured, not argued.

For the 113.67 M-parameter model with (a) seed-0 random weights and (b) the same model after N optimizer steps of the
product's own trainer (bf16 operands, fp32 masters) on the synthetic mel set of tests/brief_training.py:

  growth      : max|x_k(product) - x_k(oracle)| after k = 10, 20, 30, 40, 50 steps from the same start noise (B = 1);
  sensitivity : the PRODUCT against itself from a start noise perturbed by 1e-6 * N(0,1) — the sampler's own amplification
                of a rounding-size difference, i.e. the best any pair of fp32 implementations can agree to;
  images      : uint8 images after 50 steps: max LSB difference and fraction identical.

Writes one JSON (default gpurun_out/ddim50_parity.json); profiles/r03_ddim50_parity.md is its summary.
Usage: python tools/ddim50_parity.py [--train-steps 150] [--lr 1e-4] [--checkpoints 0,25,50,100,150] [--no-oracle]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "audio-diffusion_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

from audiodiffusion import AudioDiffusionPipeline, DDIMScheduler, Mel, UNet2DModel, _native  # noqa: E402
from brief_training import synthetic_mel  # noqa: E402

CFG256 = dict(sample_size=(256, 256), in_channels=1, out_channels=1, layers_per_block=2,
              block_out_channels=(128, 128, 256, 256, 512, 512),
              down_block_types=("DownBlock2D",) * 4 + ("AttnDownBlock2D", "DownBlock2D"),
              up_block_types=("UpBlock2D", "AttnUpBlock2D") + ("UpBlock2D",) * 4)
KS = (10, 20, 30, 40, 50)


def product_trajectory(sd, x0, dev):
    unet = UNet2DModel(**CFG256).load_state_dict(sd)
    pipe = AudioDiffusionPipeline(None, unet, Mel(), DDIMScheduler()).to(dev)
    pipe.set_progress_bar_config(disable=True)
    pipe.scheduler.set_timesteps(50)
    x, k0, snaps = x0.to(dev), 0, {}
    for k in KS:                       # the fused loop is bit-identical to the composition of its chunks (tests/test_pipeline.py)
        x, u8 = pipe._denoise(x, k0, 0.0, None, None, 0, 0, stop_step=k)
        snaps[k], k0 = x.cpu(), k
    return snaps, u8.cpu().numpy()[..., 0]


def oracle_trajectory(sd, x0):
    from oracle import schedulers as osched
    from oracle.unet import UNet2DModel as OracleUNet
    m = OracleUNet(**CFG256).eval()
    m.load_state_dict(sd)
    s = osched.DDIMScheduler()
    s.set_timesteps(50)
    x, snaps = x0.clone(), {}
    with torch.no_grad():
        for k, t in enumerate(s.timesteps):
            x = s.step(m(x, t)["sample"], t, x, eta=0.0)["prev_sample"]
            if k + 1 in KS:
                snaps[k + 1] = x.clone()
    u8 = ((x / 2 + 0.5).clamp(0, 1) * 255).round().to(torch.uint8).numpy()[:, 0]
    return snaps, u8


def measure(sd, dev, with_oracle):
    g = torch.Generator().manual_seed(1234)
    x0 = torch.randn(1, 1, 256, 256, generator=g)
    pert = x0 + 1e-6 * torch.randn(1, 1, 256, 256, generator=g)
    a, ua = product_trajectory(sd, x0, dev)
    b, ub = product_trajectory(sd, pert, dev)
    rec = {"sensitivity_product_vs_product_1e-6": {str(k): float((a[k] - b[k]).abs().max()) for k in KS},
           "sensitivity_images": {"max_lsb": int(np.abs(ua.astype(int) - ub.astype(int)).max()),
                                  "identical": float((ua == ub).mean())},
           "final_abs_max": float(a[50].abs().max()), "final_std": float(a[50].std())}
    if with_oracle:
        t0 = time.perf_counter()
        o, uo = oracle_trajectory(sd, x0)
        rec["oracle_seconds"] = round(time.perf_counter() - t0, 1)
        rec["growth_product_vs_oracle"] = {str(k): float((a[k] - o[k]).abs().max()) for k in KS}
        rec["images_product_vs_oracle"] = {"max_lsb": int(np.abs(ua.astype(int) - uo.astype(int)).max()),
                                           "identical": float((ua == uo).mean())}
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train-steps", type=int, default=150)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--checkpoints", default="0,25,50,100,150", help="optimizer-step counts at which the sampler is measured")
    ap.add_argument("--oracle-at", default="0,last", help="checkpoints that also run the 50-step CPU oracle (~45 s each)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ddim50_parity.json"))
    a = ap.parse_args()
    _native.load()
    dev = torch.device("cuda:0")
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    cps = sorted({int(c) for c in a.checkpoints.split(",") if int(c) <= a.train_steps})
    oracle_at = {cps[-1] if c == "last" else int(c) for c in a.oracle_at.split(",") if c}
    from audiodiffusion import DDPMScheduler
    from audiodiffusion import training as T
    unet = UNet2DModel(**CFG256).init_random(0)
    out = {"model": "UNet2DModel 113.67M (scripts/train_unet.py:115-137), 256x256, DDIM-50 eta=0, B=1",
           "training": {"steps": a.train_steps, "batch": a.batch, "lr": a.lr, "mixed_precision": "bf16",
                        "data": "tests/brief_training.synthetic_mel"}, "checkpoints": {}, "losses": []}
    if 0 in cps:
        out["checkpoints"]["0"] = measure(unet.state_dict(), dev, 0 in oracle_at)
        print("0", json.dumps(out["checkpoints"]["0"]), flush=True)
    flat, grads = unet.enable_training((256, 256), mixed_precision="bf16")
    opt, ns = T.AdamW(flat, lr=a.lr), DDPMScheduler()
    g = torch.Generator().manual_seed(2)
    t0 = time.perf_counter()
    for i in range(1, a.train_steps + 1):
        clean = synthetic_mel(a.batch, (256, 256), g).to(dev)
        noise = torch.randn(clean.shape, generator=g).to(dev)
        ts = torch.randint(0, 1000, (a.batch,), generator=g)
        loss = unet.train_step(ns.add_noise(clean, noise, ts), ts, noise)
        opt.step(grads, clip=T.clip_grad_norm_(grads, 1.0))
        unet.refresh_weights()
        out["losses"].append(float(loss))
        if i in cps:
            unet.sync_state_dict_from_flat()
            out["checkpoints"][str(i)] = measure(unet.state_dict(), dev, i in oracle_at)
            print(i, f"loss {float(loss):.4f}", json.dumps(out["checkpoints"][str(i)]), flush=True)
    out["train_seconds_incl_measurements"] = round(time.perf_counter() - t0, 1)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
