#!/usr/bin/env python
"""bench.py — mel-spectrograms/sec on MI355X for BASELINE.json's metric (256x256, DDIM-50).

One "step" = one pass of the hot path over one batch: a full DDIM-50 sampling (50 x {UNet2D forward + fused
scheduler epilogue}, last step emits the uint8 image) of B_gpu = 32 spectrograms of 256x256 per GPU from synthetic
Gaussian noise with seeded random-init weights of the `scripts/train_unet.py:115-137` architecture
(config 3 of BASELINE.json, per-GPU shard of its batch 256 on 8 GPUs). Inputs are resident in HBM before the timed
region. N > 1: one process per GPU (torchrun), the global noise batch is generated from one seed and row-sharded,
no collective inside the loop, one all_gather of the uint8 images at the end of each step ("scaling": "weak").

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     — dominant kernel (the exact-f32 MFMA implicit-GEMM convolution) measured LIVE with HIP events around
                 every launch of one extra eager forward on the same stream: algorithmic FLOPs / launch time;
  cpu_baseline — the CPU oracle (oracle/, a port: the reference cannot be imported without diffusers/librosa) timed on
                 the host cores on a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CFG256 = dict(sample_size=256, in_channels=1, out_channels=1, layers_per_block=2,
              block_out_channels=(128, 128, 256, 256, 512, 512),
              down_block_types=("DownBlock2D",) * 4 + ("AttnDownBlock2D", "DownBlock2D"),
              up_block_types=("UpBlock2D", "AttnUpBlock2D") + ("UpBlock2D",) * 4)
F1_TFLOP = 0.496          # algorithmic TFLOP per UNet forward per sample @256^2
A1_GB, W_GB = 1.871, 0.4547  # fused-minimum activation bytes per forward per sample; weight bytes per forward per GPU
PEAK_F32_TF = 157.3       # MI355X dense fp32 MFMA == vector peak (/opt/skills/guides/MI355X_MICROARCH.md)
PEAK_HBM_TBS = 8.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--batch-per-gpu", type=int, default=32)
    p.add_argument("--ddim-steps", type=int, default=50)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--mode", choices=["sample", "train"], default="sample",
                   help="sample (default): BASELINE.json's metric. train: config 5 (scripts/train_unet.py step, fp32).")
    p.add_argument("--train-batch-per-gpu", type=int, default=16)
    p.add_argument("--mixed-precision", choices=["no", "bf16"], default="no",
                   help="train mode only: bf16 = BASELINE.json config 5 as written (bf16 MFMA operands, fp32 accumulate)")
    return p.parse_args()


def train_main(a, world, rank, dev):
    """Extra (non-default) leg: one step = fused add_noise + native UNet forward+backward + bucketed gradient all-reduce +
    clip + fused AdamW/EMA + weight re-pack at 256x256, fp32, batch 16 per GPU, synthetic data (BASELINE.json config 5)."""
    from audiodiffusion import DDPMScheduler, UNet2DModel
    from audiodiffusion import training as T
    B = a.train_batch_per_gpu
    unet = UNet2DModel(**CFG256).init_random(0)
    flat, grads = unet.enable_training(mixed_precision=a.mixed_precision)
    opt, ema, red = T.AdamW(flat), T.EMAModel(flat), T.GradAllReducer(grads)
    if world > 1:
        red.attach(unet)          # gradient buckets are all-reduced (RCCL) from inside the reverse pass
    sched = DDPMScheduler()
    g = torch.Generator().manual_seed(7 + rank)
    clean = (torch.rand(B, 1, 256, 256, generator=g) * 2 - 1).to(dev)
    noise = torch.randn(B, 1, 256, 256, generator=g).to(dev)
    ts = torch.randint(0, 1000, (B,), generator=g)

    def step():
        noisy = sched.add_noise(clean, noise, ts)
        red.begin_step()
        loss = unet.train_step(noisy, ts, noise)
        red.start(), red.finish()
        opt.step(grads, clip=T.clip_grad_norm_(grads, 1.0), ema=ema, ema_decay=ema.next_decay())
        unet.refresh_weights()
        return loss

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    sync()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    if rank == 0:
        elapsed = float(el.item())
        value = world * B * a.steps / elapsed
        print(json.dumps({
            "metric": "training samples/sec (256x256 UNet2D, fwd+bwd+AdamW+EMA)", "value": round(value, 3),
            "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if a.mixed_precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"scripts/train_unet.py step, 256x256, batch {B}/GPU, " +
                                   ("--mixed_precision bf16 (bf16 MFMA operands on the 3x3 convolutions, fp32 accumulate/storage)"
                                    if a.mixed_precision == "bf16" else "fp32 (reference default mixed_precision=no)"),
                       "global_batch": world * B, "parallelism": f"data parallel x{world}, bucketed RCCL all-reduce"},
            "TFLOPs_3x_fwd": round(value * 3 * F1_TFLOP / world, 2), "final_loss": float(loss)}), flush=True)


def cpu_baseline(sd, n_threads):
    """Oracle (port) on the host cores: 1 warm-up + 3 timed {UNet forward + DDIM step} at B=1, extrapolated x50."""
    from oracle.schedulers import DDIMScheduler
    from oracle.unet import UNet2DModel
    torch.set_num_threads(n_threads)
    m = UNet2DModel(**CFG256).eval()
    m.load_state_dict(sd)
    s = DDIMScheduler()
    s.set_timesteps(50)
    x = torch.randn(1, 1, 256, 256, generator=torch.Generator().manual_seed(42))
    times = []
    with torch.no_grad():
        for i, t in enumerate(s.timesteps[:4]):
            t0 = time.perf_counter()
            x = s.step(m(x, t)["sample"], t, x, eta=0.0)["prev_sample"]
            if i > 0:
                times.append(time.perf_counter() - t0)
    per_step = sum(times) / len(times)
    return {"value": 1.0 / (50 * per_step), "unit": "mel-spectrograms/s", "cores": n_threads, "kind": "port",
            "sample": f"3 timed {{UNet fwd + DDIM step}} of 50 at B=1, 256x256 fp32 (torch-CPU oracle, {per_step:.2f} s/step), "
                      "linear extrapolation to 50 steps"}


def roofline(unet, x, B):
    """HIP-event pair around every launch of one eager forward (csrc/unet_exec.hip adm_unet_profile)."""
    from audiodiffusion import _native as N
    cap = 1024
    recs = (N.OpProfile * cap)()
    n = C.c_int(0)
    out = torch.empty_like(x)
    best = None
    for _ in range(3):
        N.check(N.lib().adm_unet_profile(unet._ensure_handle(), N.ptr(x), 500.0, N.ptr(out), B, recs, cap, C.byref(n),
                                         N.stream_for(x)))
        rows = [(r.kind, r.variant, r.ms, r.flops, r.bytes) for r in recs[: n.value]]
        tot = sum(r[2] for r in rows)
        if best is None or tot < best[0]:
            best = (tot, rows)
    rows = best[1]
    # dominant kernel = the convolution variant with the largest share of the forward
    KERNELS = {
        4313: "adm::conv_wino3_kernel<false> (Winograd F(2x2,3x3) 3x3 stride-1 conv, v_mfma_f32_16x16x4_f32, persistent "
              "wave-specialised: 4 producer + 4 consumer waves, 64-cout x 8x16-pixel tile)",
        2314: "adm::conv_mfma_pf_kernel<3,2,2> (v_mfma_f32_32x32x2_f32 implicit GEMM, 3x3 stride 1, 128-cout tile)",
    }
    per_var = {}
    for r in rows:
        if r[0] == 1:
            e = per_var.setdefault(r[1], [0.0, 0.0, 0])
            e[0] += r[2]; e[1] += r[3]; e[2] += 1
    var = max(per_var, key=lambda v: per_var[v][0])
    ms, fl, cnt = per_var[var]
    by_kind = {}
    for k, v, t, f, b in rows:
        e = by_kind.setdefault(k, [0, 0.0, 0.0, 0.0])
        e[0] += 1; e[1] += t; e[2] += f; e[3] += b
    names = {0: "groupnorm_stats", 1: "conv_mfma", 2: "attention", 3: "conv_small", 4: "temb_proj"}
    breakdown = {names[k]: {"launches": e[0], "ms": round(e[1], 3), "TFLOP/s": round(e[2] / e[1] / 1e9, 2) if e[1] else None,
                            "GB/s": round(e[3] / e[1] / 1e6, 1) if e[1] else None} for k, e in by_kind.items()}
    breakdown["conv_by_variant"] = {str(v): {"launches": e[2], "ms": round(e[0], 3), "TFLOP/s": round(e[1] / e[0] / 1e9, 2)}
                                    for v, e in sorted(per_var.items())}
    ach = fl / (ms * 1e-3) / 1e12
    out = {"bound": "mfma", "kernel": KERNELS.get(var, f"conv variant {var}"),
           "achieved": round(ach, 2), "peak": PEAK_F32_TF, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_TF, 4),
           "traffic": None, "launches_per_forward": cnt, "avg_launch_us": round(ms / cnt * 1e3, 2),
           "avg_flops_per_launch": fl / cnt, "share_of_forward_time": round(ms / sum(r[2] for r in rows), 3),
           "forward_breakdown": breakdown}
    # HBM bytes per launch of the dominant kernel: bench.py cannot collect PMC counters itself, so it reports the committed
    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass over one forward of this workload (tools/pmc_forward.py), if present.
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_forward.json")))["by_variant"].get(str(var))
    except (OSError, ValueError, KeyError):
        pmc = None
    if pmc and pmc.get("launches_per_forward") == cnt:
        by = sum(r[4] for r in rows if r[0] == 1 and r[1] == var)
        out["traffic"] = round(pmc["hbm_bytes_per_launch"])
        out["traffic_unit"] = "B/launch"
        out["algorithmic_bytes_per_launch"] = round(by / cnt)
        out["traffic_source"] = ("profiles/r01_pmc_forward.json: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate "
                                 "passes over one B=32 forward of this workload; includes Infinity-Cache hits")
    if var // 100 == 43:
        # `achieved` is ALGORITHMIC (direct-convolution) FLOPs per second, the contract's definition; the Winograd kernel
        # executes 16/36 of them on the matrix pipe, so the pipe utilisation is reported separately.
        out["executed_mfma_TFLOPs"] = round(ach / 2.25, 2)
        out["mfma_util"] = round(ach / 2.25 / PEAK_F32_TF, 4)
        out["note"] = ("achieved/frac use algorithmic direct-conv FLOPs (2*Cout*Cin*9*H*W*N per launch); Winograd F(2x2,3x3) "
                       "executes 1/2.25 of them as MFMA work (mfma_util), so frac may exceed 1")
    return out


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a MI355X (the hot path has no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI

    if a.mode == "train":
        train_main(a, world, rank, dev)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    from audiodiffusion import AudioDiffusionPipeline, DDIMScheduler, Mel, UNet2DModel
    unet = UNet2DModel(**CFG256).init_random(0)          # identical seeded weights on every rank
    pipe = AudioDiffusionPipeline(None, unet, Mel(), DDIMScheduler()).to(dev)
    pipe.set_progress_bar_config(disable=True)
    B = a.batch_per_gpu
    # global noise from one seed, rank r takes rows [r*B, (r+1)*B): the result does not depend on the GPU count
    g = torch.Generator().manual_seed(42)
    noise = torch.randn(world * B, 1, 256, 256, generator=g)[rank * B:(rank + 1) * B].contiguous().to(dev)
    gathered = torch.empty((world * B, 256, 256, 1), dtype=torch.uint8, device=dev) if world > 1 else None

    def step():
        _, u8 = pipe._denoise(noise, 0, 0.0, None, None, 0, 0, use_graph=not a.no_graph)
        if world > 1:
            dist.all_gather_into_tensor(gathered, u8)
        return u8

    pipe.scheduler.set_timesteps(a.ddim_steps)
    for _ in range(a.warmup):
        step()

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    if rank == 0:
        value = world * B * a.steps / elapsed
        fwd_per_s = value * a.ddim_steps
        res = {
            "metric": "mel-spectrograms/sec (256x256, DDIM-50)", "value": round(value, 4), "unit": "mel-spectrograms/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"teticio/audio-diffusion-ddim-256 architecture (UNet2DModel 113.67M params, random init seed 0), "
                                   f"DDIM-{a.ddim_steps} eta=0, 256x256, batch {B}/GPU (config 3 per-GPU shard), noise -> uint8 image",
                       "global_batch": world * B, "ddim_steps": a.ddim_steps, "hipgraph": not a.no_graph,
                       "parallelism": f"batch-shard x{world}, no in-loop collective"},
            "whole_loop": {"fp32_TFLOPs": round(fwd_per_s * F1_TFLOP / world, 2),
                           "fp32_frac_of_157.3": round(fwd_per_s * F1_TFLOP / world / PEAK_F32_TF, 4),
                           "hbm_frac_fused_min_bytes": round(fwd_per_s / world * (A1_GB + W_GB / B) / 1e3 / PEAK_HBM_TBS, 4)},
        }
        res["roofline"] = roofline(unet, noise, B)
        if not a.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only (torchrun pins OMP_NUM_THREADS=1 per rank)
            res["cpu_baseline"] = cpu_baseline(unet.state_dict(), torch.get_num_threads())
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
