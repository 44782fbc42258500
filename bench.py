#!/usr/bin/env python
"""bench.py — mel-spectrograms/sec on MI355X for BASELINE.json's metric (256x256, DDIM-50).

One "step" = one pass of the hot path over one batch: a full DDIM-50 sampling (50 x {UNet2D forward + fused
scheduler epilogue}, last step emits the uint8 image) of B_gpu = 32 spectrograms of 256x256 per GPU from synthetic
Gaussian noise with seeded random-init weights of the `scripts/train_unet.py:115-137` architecture
(config 3 of BASELINE.json, per-GPU shard of its batch 256 on 8 GPUs). Inputs are resident in HBM before the timed
region. N > 1: one process per GPU (torchrun), the global noise batch is generated from one seed and row-sharded,
no collective inside the loop, one all_gather of the uint8 images at the end of each step ("scaling": "weak").

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     — dominant kernel (since round 5 the Winograd F(4x4,3x3) fp32-MFMA convolution) measured LIVE with HIP events
                 around every launch of one extra eager forward on the same stream. `achieved`/`frac` are the EXECUTED
                 matrix-pipe FLOP/s against the 157.3 TF fp32 MFMA peak (a real roofline fraction, <= 1); the algorithmic
                 (direct-convolution) rate, which Winograd makes 4x (F(2x2): 2.25x) larger, is reported beside it;
  cpu_baseline — the CPU oracle (oracle/, a port: the reference cannot be imported without diffusers/librosa) timed on
                 the host cores on a bounded sample of the same workload;
  train        — BASELINE.json config 5 beside it: 10 optimizer steps of scripts/train_unet.py's step at 256x256,
                 batch 16 per GPU, --mixed_precision bf16 (forward + backward + bucketed gradient all-reduce over RCCL +
                 clip + AdamW/EMA + weight re-pack); samples/s over all ranks;
  mel          — the audio codec either side of the loop (Mel.audio_slice_to_image / Mel.image_to_audio, batched),
                 clips/s and algorithmic GB/s on device-resident buffers.

Test hooks (never set by the driver): ADM_BENCH_EMU=1 runs the same code on the CPU-emulation build of the kernels over
gloo at toy sizes (tests/test_bench_script.py launches this very file with 2 ranks); ADM_BENCH_FORCE_PG=1 builds the
process group and runs the collectives even at world size 1 (a one-GPU shake-out of the RCCL calls).
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
import threading
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
CFG_TOY = dict(sample_size=16, in_channels=1, out_channels=1, layers_per_block=1, block_out_channels=(32, 64),
               down_block_types=("DownBlock2D", "AttnDownBlock2D"), up_block_types=("AttnUpBlock2D", "UpBlock2D"))
F1_TFLOP = 0.496          # algorithmic TFLOP per UNet forward per sample @256^2
A1_GB, W_GB = 1.871, 0.4547  # fused-minimum activation bytes per forward per sample; weight bytes per forward per GPU
CFG_COND = dict(sample_size=(64, 64), in_channels=1, out_channels=1, layers_per_block=2, block_out_channels=(128, 256, 512, 512),
                down_block_types=("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",),
                up_block_types=("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3, cross_attention_dim=100, attention_head_dim=8)
PEAK_F32_TF = 157.3       # MI355X dense fp32 MFMA == vector peak (/opt/skills/guides/MI355X_MICROARCH.md)
PEAK_BF16_TF = 2500.0     # dense bf16 MFMA
PEAK_HBM_TBS = 8.0
MEL_CLIP_MB = 0.59        # algorithmic bytes per clip of the Mel codec (4 B/sample + 1 B/pixel)
EMU = os.environ.get("ADM_BENCH_EMU") == "1"            # test hook: CPU emulation build + gloo, toy sizes
FORCE_PG = os.environ.get("ADM_BENCH_FORCE_PG") == "1"  # test hook: process group + collectives even at world size 1


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--batch-per-gpu", type=int, default=32)
    p.add_argument("--ddim-steps", type=int, default=50)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--no-train-leg", action="store_true", help="skip the config-5 training sub-record")
    p.add_argument("--no-mel-leg", action="store_true", help="skip the Mel codec sub-record")
    p.add_argument("--no-configs-leg", action="store_true", help="skip the BASELINE configs 2 / 4 sub-record")
    p.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                   help="weak (default): --batch-per-gpu spectrograms on every GPU. strong: --global-batch (config 3 as written: "
                        "256) split over the GPUs")
    p.add_argument("--global-batch", type=int, default=256, help="global batch of --scaling strong")
    p.add_argument("--mode", choices=["sample", "train"], default="sample",
                   help="sample (default): BASELINE.json's metric. train: config 5 (scripts/train_unet.py step) as the main line.")
    p.add_argument("--train-batch-per-gpu", type=int, default=16)
    p.add_argument("--train-steps", type=int, default=10)
    p.add_argument("--train-leg-timeout", type=float, default=300.0,
                   help="seconds after which a hung training leg is reported as a failure beside the measured headline")
    p.add_argument("--mixed-precision", choices=["no", "bf16", "fp16"], default="bf16",
                   help="training precision: bf16 = BASELINE.json config 5 as written (bf16 MFMA operands, fp32 accumulate)")
    return p.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks ourselves (one process per GPU
    under torch.distributed.run, rendezvous on 127.0.0.1) and hand their output through — rank 0 prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd, env=env).returncode)


class Job:
    """Rank / device / process-group plumbing shared by every leg (one process per GPU)."""

    def __init__(self, a):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        assert self.world == a.gpus, f"--gpus {a.gpus} but the launcher's WORLD_SIZE is {self.world}"
        if EMU:
            from audiodiffusion import _native
            _native.load(os.path.join(ROOT, "tests", "emu", "libadm_emu.so"))
            self.dev = torch.device("cpu")
        else:
            assert torch.cuda.is_available(), "bench.py needs a MI355X (the hot path has no CPU fallback)"
            torch.cuda.set_device(self.local)
            self.dev = torch.device("cuda", self.local)
        self.pg = self.world > 1 or FORCE_PG
        if self.pg:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29512")
            if EMU:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)  # RCCL over xGMI

    def sync(self):
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)
        if self.pg:
            dist.barrier()
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)

    def timed(self, step, steps, warmup):
        """`warmup` untimed steps, then EXACTLY `steps` steps between barrier + synchronize on both sides; MAX over ranks."""
        for _ in range(warmup):
            step()
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.sync()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=self.dev)
        if self.pg:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item())

    def close(self):
        if self.pg:
            dist.barrier()
            dist.destroy_process_group()


def train_leg(job, B, steps, warmup, mixed_precision):
    """scripts/train_unet.py's step (:226-267) on synthetic data: fused add_noise, native UNet forward + backward with the
    bucketed gradient all-reduce issued from inside the reverse pass, clip, fused AdamW + EMA, weight re-pack."""
    from audiodiffusion import DDPMScheduler, UNet2DModel
    from audiodiffusion import training as T
    cfg = CFG_TOY if EMU else CFG256
    hw = cfg["sample_size"]
    mp = "no" if EMU else mixed_precision
    unet = UNet2DModel(**cfg).init_random(0)
    flat, grads = unet.enable_training(mixed_precision=mp)
    opt, ema = T.AdamW(flat), T.EMAModel(flat)
    # At N = 1 the driver's run has no process group. The HEADLINE of this leg is then measured without one (comparable with every earlier
    # round's N = 1 figure — ADVICE r5); afterwards the leg builds a ONE-RANK group of its own and times a second, shorter run in it, so that
    # the bucket hook -> asynchronous RCCL all-reduce path executes under a real backward pass on every bench run, reported
    # as `one_rank_group` beside the headline.
    red = T.GradAllReducer(grads, force=FORCE_PG)
    if job.pg:
        red.attach(unet)          # gradient buckets are all-reduced (RCCL) from inside the reverse pass
    sched = DDPMScheduler()
    g = torch.Generator().manual_seed(7 + job.rank)
    clean = (torch.rand(B, 1, hw, hw, generator=g) * 2 - 1).to(job.dev)
    noise = torch.randn(B, 1, hw, hw, generator=g).to(job.dev)
    ts = torch.randint(0, 1000, (B,), generator=g)
    last = {}

    scaler = T.GradScaler() if mp == "fp16" else None

    def step():
        noisy = sched.add_noise(clean, noise, ts)
        red.begin_step()
        last["loss"] = unet.train_step(noisy, ts, noise, loss_scale=scaler.get_scale() if scaler else 1.0)
        last["overlapped"] = red.overlapped
        red.start(), red.finish()
        if scaler:                      # GradScaler: un-scale + clip in one pass, skip the step on an overflow
            clip, found_inf = scaler.unscale_and_clip_(grads, 1.0)
            scaler.update(found_inf)
            if found_inf:
                return
        else:
            clip = T.clip_grad_norm_(grads, 1.0)
        opt.step(grads, clip=clip, ema=ema, ema_decay=ema.next_decay())
        unet.refresh_weights()

    elapsed = job.timed(step, steps, warmup)
    value = job.world * B * steps / elapsed
    # every bucket of the last step was queued from inside the reverse pass — on EVERY rank (MIN over the job)
    every = None
    if job.pg:
        ok = torch.tensor([1 if last.get("overlapped", 0) == len(red.bounds) else 0], dtype=torch.int32, device=job.dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        every = bool(int(ok.item()))
    own = None
    if not job.pg and not dist.is_initialized() and os.environ.get("ADM_BENCH_TRAIN_PG", "1") == "1":
        try:
            import socket
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            with socket.socket() as sk:              # a free port, asked of the kernel (not pid % 300: concurrent runs collided)
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            if EMU:
                dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
            else:
                dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=job.dev)
            try:
                red = T.GradAllReducer(grads, force=True)
                red.attach(unet)
                n2 = max(3, steps // 2)
                el2 = job.timed(step, n2, 1)
                own = {"ms_per_step": round(el2 / n2 * 1e3, 2), "steps": n2, "allreduce_buckets": len(red.bounds),
                       "allreduce_buckets_overlapped": last.get("overlapped", 0),
                       "backend": "gloo" if EMU else "nccl (RCCL), world_size 1"}
            finally:
                dist.destroy_process_group()
        except Exception as ex:  # noqa: BLE001 — the headline above stands; the record says why the second run is missing
            own = {"error": f"{type(ex).__name__}: {ex}"[:200]}
    return {"allreduce_overlapped_on_every_rank": every, "one_rank_group": own,
            "metric": "training samples/sec (256x256 UNet2D, fwd+bwd+all-reduce+AdamW+EMA)", "value": round(value, 3),
            "unit": "samples/s", "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 2),
            "batch_per_gpu": B, "global_batch": job.world * B, "dtype": mp if mp in ("bf16", "fp16") else "f32",
            "workload": "scripts/train_unet.py step, 256x256, " +
                        (f"--mixed_precision {mp} ({mp} MFMA operands on the 3x3 and 1x1 convolutions of all three passes, fp32 "
                         "accumulate / storage / optimizer)" if mp != "no" else "fp32 (reference default mixed_precision=no)"),
            "parallelism": f"data parallel x{job.world}, 25 MB gradient buckets all-reduced from inside the reverse pass",
            "allreduce_buckets": len(red.bounds), "allreduce_buckets_overlapped": last.get("overlapped", 0),
            "TFLOPs_3x_fwd": round(value * 3 * F1_TFLOP / job.world, 2),
            ("frac_of_16bit_mfma_peak" if mp != "no" else "frac_of_fp32_peak"):
                round(value * 3 * F1_TFLOP / job.world / (PEAK_BF16_TF if mp != "no" else PEAK_F32_TF), 4),
            "final_loss": float(last["loss"])}


def mel_leg(job, n_fwd=256, n_inv=32):
    """Mel codec on device-resident buffers, batched (the reference converts clip by clip on one host core,
    pipeline_audio_diffusion.py:135-141,201): forward = audio -> uint8 log-mel image, inverse = image -> audio
    (NNLS start point + 32 Griffin-Lim iterations)."""
    from audiodiffusion import Mel
    from audiodiffusion import _native as N
    mel = Mel(x_res=16, y_res=16, hop_length=64, n_fft=256, n_iter=2) if EMU else Mel()
    if EMU:
        n_fwd, n_inv = 4, 2
    h = mel._ensure_handle()
    n = mel.slice_size
    g = torch.Generator().manual_seed(3)
    audio = (0.3 * torch.randn(n_fwd, n, generator=g)).to(job.dev)
    frames = 1 + n // mel.hop_length
    img = torch.empty((n_fwd, mel.n_mels, frames), dtype=torch.uint8, device=job.dev)
    st = N.stream_for(audio)

    def fwd():
        N.check(N.lib().adm_mel_forward(h, N.ptr(audio), 0, n_fwd, n, n, N.ptr(img), st))

    def wall(fn, reps):
        fn()
        if job.dev.type == "cuda":
            torch.cuda.synchronize(job.dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        if job.dev.type == "cuda":
            torch.cuda.synchronize(job.dev)
        return (time.perf_counter() - t0) / reps

    t_f = wall(fwd, 1 if EMU else 10)
    n_bins = 1 + mel.n_fft // 2
    images = img[:n_inv].contiguous()
    phase = torch.rand((n_inv, n_bins, frames), dtype=torch.float64, device=job.dev)
    out = torch.empty((n_inv, mel.hop_length * (frames - 1)), dtype=torch.float32, device=job.dev)

    def inv():
        N.check(N.lib().adm_mel_inverse(h, N.ptr(images), N.ptr(phase), n_inv, frames, N.ptr(out), None, None, st))

    t_i = wall(inv, 1 if EMU else 3)
    # The bound that applies is the fp64 VECTOR pipe, not HBM: numpy runs these FFTs in double and so do the kernels
    # (fp32 would move quiet dB bins by 0.02-0.05 dB and break the >= 99.9 %-identical image bar). One wave transforms one 2048-sample frame
    # in ~1150 fp64 wave-instructions (real-input trick, 1024 complex points, radix 16 x 16 x 4; k_mel.hip) at 4 cycles each on a SIMD.
    def fp64_floor_ms(frame_transforms):
        return frame_transforms * 1150 * 4 / (256 * 4 * 2.4e9) * 1e3
    fl_f = fp64_floor_ms(n_fwd * frames)
    fl_i = fp64_floor_ms(n_inv * frames * 2 * mel.n_iter)          # Griffin-Lim: one inverse + one forward transform per frame and iteration
    return {"forward": {"clips_per_s": round(n_fwd / t_f, 1), "batch": n_fwd, "ms": round(t_f * 1e3, 3),
                        "bound": "fp64 VALU (FFT in double, as numpy)", "fp64_valu_floor_ms": round(fl_f, 4),
                        "frac_of_fp64_valu_floor": round(fl_f / (t_f * 1e3), 4),
                        "GB/s_algorithmic": round(n_fwd * MEL_CLIP_MB / 1e3 / t_f, 1),
                        "hbm_frac_informational": round(n_fwd * MEL_CLIP_MB / 1e6 / t_f / PEAK_HBM_TBS, 4)},
            "inverse": {"clips_per_s": round(n_inv / t_i, 1), "batch": n_inv, "ms": round(t_i * 1e3, 3),
                        "griffin_lim_iters": mel.n_iter,
                        "bound": "fp64 VALU (2 FFTs per frame and Griffin-Lim iteration) + LDS exchanges", "fp64_valu_floor_ms": round(fl_i, 4),
                        "frac_of_fp64_valu_floor": round(fl_i / (t_i * 1e3), 4),
                        "GB/s_algorithmic": round(n_inv * MEL_CLIP_MB / 1e3 / t_i, 2)},
            "config": "toy" if EMU else "x_res 256, y_res 256, n_fft 2048, hop 512, 22050 Hz (5.9 s clips)"}


def cpu_baseline(sd, cores):
    """Oracle (port) on the host cores, a bounded sample (~15-25 s) of the same workload: {UNet forward + DDIM step} at B=1,
    256x256 fp32. One warm-up + one probe step at up to three thread counts (torch-CPU convolutions stop scaling well before the
    box's 256 threads), then 10 consecutive steps of the DDIM-50 schedule at the fastest count; linear extrapolation to 50."""
    from oracle.schedulers import DDIMScheduler
    from oracle.unet import UNet2DModel
    m = UNet2DModel(**CFG256).eval()
    m.load_state_dict(sd)
    s = DDIMScheduler()
    s.set_timesteps(50)
    x = torch.randn(1, 1, 256, 256, generator=torch.Generator().manual_seed(42))

    def run(ts):
        nonlocal x
        times = []
        with torch.no_grad():
            for t in ts:
                t0 = time.perf_counter()
                x = s.step(m(x, t)["sample"], t, x, eta=0.0)["prev_sample"]
                times.append(time.perf_counter() - t0)
        return times

    tried = {}
    for nt in sorted({min(16, cores), min(32, cores), min(64, cores)}):   # all 256 hardware threads of the box: 87 s per step
        torch.set_num_threads(nt)
        tried[nt] = run(s.timesteps[:2])[1]
    nt = min(tried, key=tried.get)
    torch.set_num_threads(nt)
    full = tried[nt] * 50 <= 60.0          # the complete 50-step sampling when it fits a minute of CPU work
    if full:
        x = torch.randn(1, 1, 256, 256, generator=torch.Generator().manual_seed(42))
        times = run(s.timesteps)
        total = sum(times)
        return {"value": 1.0 / total, "unit": "mel-spectrograms/s", "cores": nt, "kind": "port",
                "sample": f"the COMPLETE DDIM-50 sampling at B=1, 256x256 fp32 (torch-CPU oracle: 50 x {{UNet fwd + DDIM step}}, "
                          f"{total:.1f} s on {nt} threads; thread-count probe: " + ", ".join(f"{k}: {v:.2f} s" for k, v in tried.items()) + ")"}
    times = run(s.timesteps[2:12])
    per_step = sum(times) / len(times)
    return {"value": 1.0 / (50 * per_step), "unit": "mel-spectrograms/s", "cores": nt, "kind": "port",
            "sample": f"{len(times)} consecutive {{UNet fwd + DDIM step}} of the 50 at B=1, 256x256 fp32 (torch-CPU oracle, "
                      f"{per_step:.2f} s/step on {nt} threads, {sum(times):.1f} s of CPU work; thread-count probe: " +
                      ", ".join(f"{k}: {v:.2f} s" for k, v in tried.items()) + "), linear extrapolation to 50 steps"}


def side_cpu_baselines(nt):
    """BASELINE.md §3's CPU protocol for the SIDE legs (the oracle on `nt` host threads, outside every timed region, bounded to about a
    minute in total): config 1 in full (64x64 DDPM-10, B = 1), config 4 (20 latent 32x32 UNet steps + one AutoencoderKL decode at B = 1),
    config 5 (optimizer steps at B = 2, fp32: forward + backward + clip + AdamW), and the Mel codec single-threaded as the reference
    calls it per image (pipeline_audio_diffusion.py:201; the only wall-clock claim in the reference tree is app.py:20-22)."""
    import numpy as np
    import torch.nn.functional as F
    from oracle import mel as omel
    from oracle.schedulers import DDPMScheduler
    from oracle.unet import UNet2DModel
    from oracle.vae import AutoencoderKL
    out = {}
    torch.set_num_threads(nt)

    def cfg(res):
        c = dict(CFG256)
        c["sample_size"] = res
        return c

    def ddpm_steps(m, res, n, B=1):
        s = DDPMScheduler()
        s.set_timesteps(1000 if n != 10 else 10)
        g = torch.Generator().manual_seed(5)
        x = torch.randn(B, 1, res, res, generator=g)
        t0 = time.perf_counter()
        with torch.no_grad():
            for t in s.timesteps[:n]:
                x = s.step(m(x, t)["sample"], t, x, generator=g)["prev_sample"]
        return time.perf_counter() - t0

    torch.manual_seed(0)
    m64 = UNet2DModel(**cfg(64)).eval()
    ddpm_steps(m64, 64, 1)
    t = ddpm_steps(m64, 64, 10)
    out["config_1"] = {"s_per_sample": round(t, 3), "ms_per_step": round(t * 100, 1), "cores": nt, "kind": "port",
                       "sample": "the complete 10-step 64x64 DDPM sampling at B = 1 (oracle UNet + scheduler)"}
    del m64
    m32 = UNet2DModel(**cfg(32)).eval()
    ddpm_steps(m32, 32, 1)
    t = ddpm_steps(m32, 32, 20)
    vae = AutoencoderKL(sample_size=(256, 256), in_channels=1, out_channels=1, latent_channels=1, layers_per_block=2,
                        block_out_channels=(128, 256, 512, 512), down_block_types=("DownEncoderBlock2D",) * 4,
                        up_block_types=("UpDecoderBlock2D",) * 4).eval()
    z = torch.randn(1, 1, 32, 32, generator=torch.Generator().manual_seed(6))
    t0 = time.perf_counter()
    with torch.no_grad():
        vae.decode(z / 0.18215)
    td = time.perf_counter() - t0
    out["config_4"] = {"ms_per_step": round(t / 20 * 1e3, 1), "s_vae_decode": round(td, 3), "cores": nt, "kind": "port",
                       "spectrograms_per_s_extrapolated": round(1.0 / (t / 20 * 1000 + td), 5),
                       "sample": "20 latent 32x32 DDPM steps + ONE AutoencoderKL decode to 256x256 at B = 1, x50 extrapolation of the loop"}
    del m32, vae
    try:
        from oracle.schedulers import DDIMScheduler as ODDIM
        from oracle.unet_condition import UNet2DConditionModel as OCond
        mc = OCond(**CFG_COND).eval()
        sc = ODDIM()
        sc.set_timesteps(50)
        g = torch.Generator().manual_seed(8)
        x = torch.randn(1, 1, 64, 64, generator=g)
        enc = torch.randn(1, 1, 100, generator=g)
        with torch.no_grad():
            mc(x, sc.timesteps[0], enc)
            t0 = time.perf_counter()
            for t_ in sc.timesteps[:10]:
                x = sc.step(mc(x, t_, enc)["sample"], t_, x)["prev_sample"]
        tcnd = time.perf_counter() - t0
        out["conditional"] = {"ms_per_step": round(tcnd / 10 * 1e3, 1), "spectrograms_per_s_extrapolated": round(1.0 / (tcnd * 5), 5),
                              "cores": nt, "kind": "port",
                              "sample": "10 of the 50 DDIM steps of the conditional 64x64 latent UNet at B = 1 (oracle), x5 extrapolation"}
        del mc
    except Exception as e:  # noqa: BLE001
        out["conditional"] = {"error": f"{type(e).__name__}: {e}"}
    m = UNet2DModel(**CFG256)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)
    g = torch.Generator().manual_seed(7)
    x, tgt = torch.randn(2, 1, 256, 256, generator=g), torch.randn(2, 1, 256, 256, generator=g)
    ts = torch.randint(0, 1000, (2,), generator=g)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        loss = F.mse_loss(m(x, ts)["sample"], tgt)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
        times.append(time.perf_counter() - t0)
        if sum(times) > 45:
            break
    per = min(times)
    out["train"] = {"samples_per_s": round(2 / per, 4), "s_per_step": round(per, 2), "cores": nt, "kind": "port", "dtype": "f32",
                    "sample": f"{len(times)} optimizer steps at B = 2, 256x256 fp32 (oracle forward + torch autograd + clip + AdamW), best step"}
    del m, opt
    torch.set_num_threads(1)
    om = omel.Mel()
    rng = np.random.default_rng(3)
    om.load_audio(raw_audio=(0.3 * rng.standard_normal(om.slice_size)).astype(np.float32))
    om.audio_slice_to_image(0)
    t0 = time.perf_counter()
    for _ in range(10):
        img = om.audio_slice_to_image(0)
    tf = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(3):
        om.image_to_audio(img)
    ti = (time.perf_counter() - t0) / 3
    out["mel"] = {"forward_ms_per_call": round(tf * 1e3, 2), "inverse_ms_per_call": round(ti * 1e3, 1), "cores": 1, "kind": "port",
                  "sample": "oracle Mel.audio_slice_to_image x10 and Mel.image_to_audio x3 (NNLS start point + 32 Griffin-Lim iterations), "
                            "one 5.9 s clip, single thread"}
    torch.set_num_threads(nt)
    return out


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
        4313: "adm::conv_wino3_kernel (Winograd F(2x2,3x3) 3x3 stride-1 conv, v_mfma_f32_16x16x4_f32, persistent "
              "wave-specialised: 4 producer + 4 consumer waves, 64-cout x 8x16-pixel tile)",
        4314: "adm::conv_wino4_kernel (Winograd F(2x2,3x3) 3x3 stride-1 conv, v_mfma_f32_16x16x4_f32, persistent wave-specialised: "
              "4 producer waves (patch -> GroupNorm/SiLU -> B^T d B into LDS) + 4 consumer waves (16 couts x 32 tiles each, filters "
              "L2 -> registers, lane-local A^T M A), 64-cout x 8x16-pixel tile)",
        4317: "adm::conv_wino4_kernel, split K (the single-sample rule \"single_sample\": several workgroups per tile, each a part of the input "
              "channels) + ksplit_finish(_stats)_kernel",
        4315: "adm::conv_wino5_kernel (Winograd F(2x2,3x3) 3x3 stride-1 conv, v_mfma_f32_16x16x4_f32, persistent: 128-cout x 8x16-pixel "
              "workgroup tile, every input patch transformed once per 128 couts; all 8 waves are MFMA waves (16 couts x 32 tiles x 16 "
              "points each, filters L2 -> registers, lane-local A^T M A) and share the staging (patch -> GroupNorm/SiLU -> B^T d B into "
              "LDS); the two waves of a SIMD run MFMA block and staging in antiphase)",
        4316: "adm::conv_wino6_kernel (Winograd F(4x4,3x3) 3x3 stride-1 conv, v_mfma_f32_16x16x4_f32, persistent: 128-cout x 16x16-pixel "
              "workgroup tile = 16 Winograd tiles x 36 points; all 8 waves are MFMA waves (144 accumulators each, filters L2 -> registers "
              "through a ring, lane-local A^T M A) and share the staging (18x18 patch -> GroupNorm/SiLU -> B^T d B into LDS))",
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
    breakdown = {names.get(k, str(k)): {"launches": e[0], "ms": round(e[1], 3),
                                        "TFLOP/s": round(e[2] / e[1] / 1e9, 2) if e[1] else None,
                                        "GB/s": round(e[3] / e[1] / 1e6, 1) if e[1] else None} for k, e in by_kind.items()}
    breakdown["conv_by_variant"] = {str(v): {"launches": e[2], "ms": round(e[0], 3), "TFLOP/s": round(e[1] / e[0] / 1e9, 2)}
                                    for v, e in sorted(per_var.items())}
    alg = fl / (ms * 1e-3) / 1e12                       # algorithmic (direct-convolution) TFLOP/s of the dominant kernel
    wino = var // 100 == 43
    # F(2x2,3x3): 16 MFMA products per 4 outputs instead of 36 (/ 2.25); F(4x4,3x3): 36 per 16 outputs instead of 144 (/ 4)
    executed = alg / 4.0 if var == 4316 else (alg / 2.25 if wino else alg)
    total_ms = sum(r[2] for r in rows)
    out = {"bound": "mfma", "kernel": KERNELS.get(var, f"conv variant {var}"),
           "achieved": round(executed, 2), "peak": PEAK_F32_TF, "unit": "TFLOP/s", "frac": round(executed / PEAK_F32_TF, 4),
           "achieved_definition": "EXECUTED fp32 MFMA FLOPs per launch (algorithmic direct-convolution FLOPs / 4 for the F(4x4,3x3) "
                                  "Winograd kernel, / 2.25 for the F(2x2,3x3) ones) / average launch duration, HIP events on the launch stream",
           "algorithmic_TFLOPs": round(alg, 2), "algorithmic_frac_of_peak": round(alg / PEAK_F32_TF, 4),
           "traffic": None, "launches_per_forward": cnt, "avg_launch_us": round(ms / cnt * 1e3, 2),
           "avg_algorithmic_flops_per_launch": fl / cnt, "share_of_forward_time": round(ms / total_ms, 3),
           "forward_ms": round(total_ms, 3), "forward_breakdown": breakdown}
    # HBM bytes per launch of the dominant kernel: bench.py cannot collect PMC counters itself, so it reports the committed rocprofv3 --pmc
    # FETCH_SIZE / WRITE_SIZE pass over one forward of this workload (tools/pmc_forward.sh) — but ONLY a pass that was made of THIS build:
    # the file carries the library's adm_version() and a hash of the Winograd kernel sources (kernel_stamp), and a pass whose stamp differs
    # from the sources in the tree is refused (a kernel change that keeps the launch count would otherwise print stale traffic — VERDICT r5).
    stamp = kernel_stamp()
    pmc, src, stale = None, None, []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_forward.json")), reverse=True):
        try:
            doc = json.load(open(f))
            cand = doc["by_variant"].get(str(var))
        except (OSError, ValueError, KeyError):
            continue
        if not cand or cand.get("launches_per_forward") != cnt:
            continue
        if doc.get("stamp") != stamp:
            stale.append(os.path.relpath(f, ROOT))
            continue
        pmc, src = cand, os.path.relpath(f, ROOT)
        break
    if pmc:
        by = sum(r[4] for r in rows if r[0] == 1 and r[1] == var)
        out["traffic"] = round(pmc["hbm_bytes_per_launch"])
        out["traffic_unit"] = "B/launch"
        out["algorithmic_bytes_per_launch"] = round(by / cnt)
        out["traffic_source"] = (f"{src} (stamp {stamp['kernel_sources_sha16']}, adm_version {stamp['adm_version']} = this build): rocprofv3 --pmc "
                                 "FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes over one B=32 forward of this workload; includes Infinity-Cache hits")
    else:
        out["traffic_source"] = ("no PMC pass of THIS build under profiles/ (tools/pmc_forward.sh writes one; passes of other builds are refused" +
                                 (": " + ", ".join(stale[:3]) if stale else "") + ")")
    return out


def kernel_stamp():
    """What a PMC pass must have been made of to be quoted beside this build's timings: the library's ABI version and a hash of the Winograd
    kernel sources (the dominant kernel's translation units)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "audio-diffusion_amd", "csrc")
    for name in ("k_conv_wino.h", "k_conv_wino.hip", "k_conv_wino_f2.hip", "k_conv_wino_f4.hip"):
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(f.read())
    from audiodiffusion import _native as N
    return {"adm_version": int(N.lib().adm_version()), "kernel_sources_sha16": h.hexdigest()[:16]}


def configs_leg(job):
    """BASELINE.json configs 1, 2 and 4 on this GPU, bounded: config 1 in full; 50 CONSECUTIVE steps of their 1000-step DDPM schedules with
    injected per-step noise through the same native loop (per-step cost is constant, so x20 is the full sampling), config 4
    plus its AutoencoderKL decode. Builder-side probes of the full 1000 steps: tools/config_probe.py."""
    from audiodiffusion import AudioDiffusionPipeline, DDPMScheduler, Mel, UNet2DModel
    from audiodiffusion.vae import AutoencoderKL
    dev, B, n = job.dev, 16, 50
    out = {}

    def run(pipe, res, decode):
        pipe.set_progress_bar_config(disable=True)
        pipe.scheduler.set_timesteps(1000)
        g = torch.Generator().manual_seed(5)
        x = torch.randn(B, 1, res, res, generator=g).to(dev)
        step_noise = torch.randn(n, B, 1, res, res, generator=g).to(dev)

        def loop():
            return pipe._denoise(x, 0, 0.0, None, None, 0, 0, step_noise=step_noise, stop_step=n, want_u8=not decode)[0]

        def dec(lat):
            from audiodiffusion import ops
            return ops.dequant_u8(pipe.vqvae.decode(lat, _in_scale=1 / 0.18215)["sample"])

        def wall(fn, *a):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            r = fn(*a)
            torch.cuda.synchronize(dev)
            return time.perf_counter() - t0, r
        lat = loop()
        if decode:
            dec(lat)
        t_loop, lat = wall(loop)
        t_dec = wall(dec, lat)[0] if decode else 0.0
        return t_loop, t_dec

    def cfg(res):
        c = dict(CFG256)
        c["sample_size"] = res
        return c

    # config 1 (the reference's own CPU-runnable case): 64x64, DDPM, ONE sample, the complete 10-step sampling
    p1 = AudioDiffusionPipeline(None, UNet2DModel(**cfg(64)).init_random(0), Mel(x_res=64, y_res=64, hop_length=1024), DDPMScheduler()).to(dev)
    p1.set_progress_bar_config(disable=True)
    p1.unet.set_option("single_sample", 1)      # ONE sample per call: the model's single-sample rule, as AudioDiffusion selects it (include/adm.h)
    n1 = torch.randn(1, 1, 64, 64, generator=torch.Generator().manual_seed(6)).to(dev)
    p1(batch_size=1, steps=10, noise=n1.clone(), audio=False)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    p1(batch_size=1, steps=10, noise=n1.clone(), audio=False)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter() - t0
    out["config_1"] = {"workload": "audio-diffusion-64 architecture, 64x64 DDPM, 1 sample, 10 steps (complete sampling through the public "
                                   "__call__, noise -> PIL image; the model carries the single-sample rule UNet2DModel.set_option("
                                   "'single_sample', 1), as the AudioDiffusion front end sets it)", "ms_per_sample": round(t1 * 1e3, 2), "ms_per_step": round(t1 * 1e2, 3)}
    del p1
    t, _ = run(AudioDiffusionPipeline(None, UNet2DModel(**cfg(256)).init_random(0), Mel(), DDPMScheduler()).to(dev), 256, False)
    out["config_2"] = {"workload": "teticio/audio-diffusion-256 architecture, pixel-space DDPM on the 1000-step schedule, 256x256, "
                                   f"batch {B}: {n} consecutive steps (t = 999...{1000 - n}) with injected noise, timed; x{1000 // n} "
                                   "extrapolation to the full sampling",
                       "ms_per_step": round(t / n * 1e3, 3), "steps_timed": n,
                       "spectrograms_per_s_extrapolated": round(B / (t / n * 1000), 4)}
    vae = AutoencoderKL(sample_size=(256, 256), in_channels=1, out_channels=1, latent_channels=1, layers_per_block=2,
                        block_out_channels=(128, 256, 512, 512), down_block_types=("DownEncoderBlock2D",) * 4,
                        up_block_types=("UpDecoderBlock2D",) * 4).init_random(0)
    t, td = run(AudioDiffusionPipeline(vae, UNet2DModel(**cfg(32)).init_random(1), Mel(), DDPMScheduler()).to(dev), 32, True)
    out["config_4"] = {"workload": "teticio/latent-audio-diffusion-256 architecture: latent 32x32 UNet2D DDPM on the 1000-step "
                                   f"schedule + AutoencoderKL decode to 256x256, batch {B}: {n} consecutive steps and ONE decode, timed "
                                   f"separately; x{1000 // n} extrapolation of the loop + the decode = the full sampling",
                       "ms_per_step": round(t / n * 1e3, 3), "steps_timed": n, "ms_vae_decode": round(td * 1e3, 2),
                       "spectrograms_per_s_extrapolated": round(B / (t / n * 1000 + td), 4)}
    # (f2) conditional generation at size: UNet2DConditionModel as scripts/train_unet.py:139-159 builds it for a 512-resolution latent model
    # (64x64 latents, (128, 256, 512, 512), three cross-attention blocks each way, encoding = one 100-wide row per sample as
    # audiodiffusion/audio_encoder.py produces it), DDIM steps with the encoding held constant over the loop (pipeline...py:160-161)
    try:
        from audiodiffusion import DDIMScheduler, UNet2DConditionModel
        cu = UNet2DConditionModel(**CFG_COND).init_random(0)
        pc = AudioDiffusionPipeline(None, cu, Mel(x_res=64, y_res=64), DDIMScheduler()).to(dev)
        pc.set_progress_bar_config(disable=True)
        pc.scheduler.set_timesteps(50)
        g = torch.Generator().manual_seed(8)
        xc = torch.randn(B, 1, 64, 64, generator=g).to(dev)
        enc = torch.randn(B, 1, 100, generator=g).to(dev)

        def loopc():
            return pc._denoise(xc, 0, 0.0, None, None, 0, 0, encoding=enc)[0]
        loopc()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        loopc()
        torch.cuda.synchronize(dev)
        tc = time.perf_counter() - t0
        out["conditional"] = {"workload": "UNet2DConditionModel (scripts/train_unet.py:139-159: (128, 256, 512, 512), CrossAttn blocks, "
                                          f"cross_attention_dim 100, 135.6 M parameters), 64x64 latents (512-resolution model), batch {B}, "
                                          "complete DDIM-50 sampling with a constant encoding through the native loop",
                              "ms_per_step": round(tc / 50 * 1e3, 3), "steps_timed": 50, "spectrograms_per_s": round(B / tc, 3)}
        del pc, cu
    except Exception as e:  # noqa: BLE001
        out["conditional"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)
    job = Job(a)
    world, rank, dev = job.world, job.rank, job.dev

    if a.mode == "train":
        rec = train_leg(job, a.train_batch_per_gpu, a.steps, a.warmup, a.mixed_precision)
        if rank == 0:
            rec.update({"n_gpus": world, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic",
                        "config": {"workload": rec.pop("workload"), "global_batch": rec["global_batch"],
                                   "parallelism": rec.pop("parallelism")}})
            print(json.dumps(rec), flush=True)
        job.close()
        return

    from audiodiffusion import AudioDiffusionPipeline, DDIMScheduler, Mel, UNet2DModel
    cfg = CFG_TOY if EMU else CFG256
    hw = cfg["sample_size"]
    unet = UNet2DModel(**cfg).init_random(0)          # identical seeded weights on every rank
    mel = Mel(x_res=16, y_res=16, hop_length=64, n_fft=256, n_iter=1) if EMU else Mel()
    pipe = AudioDiffusionPipeline(None, unet, mel, DDIMScheduler()).to(dev)
    pipe.set_progress_bar_config(disable=True)
    B = a.batch_per_gpu
    lo, hi, n_global = rank * B, (rank + 1) * B, world * B
    if a.scaling == "strong":       # config 3 as written: one global batch, split by rows — ceil-sized shards, the last one short when the
        from audiodiffusion.distributed import shard_bounds          # GPU count does not divide it (ranks beyond the rows own nothing)
        n_global = a.global_batch
        B = (n_global + world - 1) // world
        lo, hi = shard_bounds(n_global, world, rank)
    # global noise from one seed, rank r takes rows [lo, hi): the result does not depend on the GPU count
    g = torch.Generator().manual_seed(42)
    noise = torch.randn(n_global, 1, hw, hw, generator=g)[lo:hi].contiguous().to(dev)
    gathered = torch.empty((world * B, hw, hw, 1), dtype=torch.uint8, device=dev) if job.pg else None
    padded = torch.zeros((B, hw, hw, 1), dtype=torch.uint8, device=dev) if (job.pg and hi - lo < B) else None

    def step():
        u8 = None
        if hi > lo:
            _, u8 = pipe._denoise(noise, 0, 0.0, None, None, 0, 0, use_graph=not a.no_graph)
        if job.pg:
            if padded is not None:                      # a short (or empty) last shard: the gather takes equal-sized pieces
                if u8 is not None:
                    padded[: hi - lo] = u8.reshape(hi - lo, hw, hw, 1)
                dist.all_gather_into_tensor(gathered, padded)
            else:
                dist.all_gather_into_tensor(gathered, u8)
        return u8

    pipe.scheduler.set_timesteps(a.ddim_steps)
    elapsed = job.timed(step, a.steps, a.warmup)

    res = None
    if rank == 0:
        value = n_global * a.steps / elapsed
        fwd_per_s = value * a.ddim_steps
        res = {
            "metric": "mel-spectrograms/sec (256x256, DDIM-50)", "value": round(value, 4), "unit": "mel-spectrograms/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 2),
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"teticio/audio-diffusion-ddim-256 architecture (UNet2DModel 113.67M params, random init seed 0), "
                                   f"DDIM-{a.ddim_steps} eta=0, 256x256, batch {B}/GPU (config 3 per-GPU shard), noise -> uint8 image",
                       "global_batch": n_global, "ddim_steps": a.ddim_steps, "hipgraph": not a.no_graph,
                       "parallelism": f"batch-shard x{world}, no in-loop collective"},
            "whole_loop": {"fp32_algorithmic_TFLOPs": round(fwd_per_s * F1_TFLOP / world, 2),
                           "fp32_algorithmic_frac_of_157.3": round(fwd_per_s * F1_TFLOP / world / PEAK_F32_TF, 4),
                           "hbm_frac_fused_min_bytes": round(fwd_per_s / world * (A1_GB + W_GB / B) / 1e3 / PEAK_HBM_TBS, 4)},
        }
        if EMU:
            res["config"]["workload"] = "TEST HOOK ADM_BENCH_EMU=1: toy UNet on the CPU emulation build over gloo (not a measurement)"
            res["gathered_checksum"] = int(gathered[:n_global].long().sum()) if gathered is not None else None
        else:
            # side legs never cost the headline line: a failure is reported in place
            try:
                res["roofline"] = roofline(unet, noise, B)
            except Exception as e:  # noqa: BLE001
                res["roofline"] = {"error": f"{type(e).__name__}: {e}"}
            if not a.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only (torchrun pins OMP_NUM_THREADS=1 per rank)
                try:
                    res["cpu_baseline"] = cpu_baseline(unet.state_dict(), os.cpu_count() or torch.get_num_threads())
                except Exception as e:  # noqa: BLE001
                    res["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
            if not a.no_configs_leg:
                try:
                    res["configs"] = configs_leg(job)
                except Exception as e:  # noqa: BLE001
                    res["configs"] = {"error": f"{type(e).__name__}: {e}"}
            if not a.no_cpu_baseline and world == 1:      # the side legs' CPU figures (attached to their sub-records below)
                try:
                    nt_side = res.get("cpu_baseline", {}).get("cores") or min(16, os.cpu_count() or 1)
                    side_cpu = side_cpu_baselines(int(nt_side))
                except Exception as e:  # noqa: BLE001
                    side_cpu = {"error": f"{type(e).__name__}: {e}"}
                if "error" in side_cpu:
                    res["side_cpu_baselines"] = side_cpu
                if isinstance(res.get("configs"), dict):
                    for k in ("config_1", "config_4", "conditional"):
                        if k in side_cpu and k in res["configs"]:
                            res["configs"][k]["cpu_baseline"] = side_cpu[k]
                res["_side_cpu"] = side_cpu
        if not a.no_mel_leg:
            try:
                res["mel"] = mel_leg(job)
            except Exception as e:  # noqa: BLE001 - a failing side leg must not cost the headline line
                res["mel"] = {"error": f"{type(e).__name__}: {e}"}
            if isinstance(res.get("_side_cpu"), dict) and "mel" in res["_side_cpu"]:
                res["mel"]["cpu_baseline"] = res["_side_cpu"]["mel"]
    del pipe
    # config 5 beside the headline: every rank takes part (the gradient all-reduce is a collective)
    if not a.no_train_leg:
        # The headline is measured; a side leg that HANGS (a collective that never completes on some rank) must not cost it:
        # every rank arms the same watchdog, rank 0 prints the line with the failure in place, all ranks leave.
        def give_up():
            if rank == 0:
                res["train"] = {"error": f"timeout: the training leg did not finish within {a.train_leg_timeout:.0f} s"}
                print(json.dumps(res), flush=True)
            os._exit(0)
        dog = threading.Timer(a.train_leg_timeout, give_up)
        dog.daemon = True
        dog.start()
        job.sync()
        try:
            tr = train_leg(job, 2 if EMU else a.train_batch_per_gpu, 2 if EMU else a.train_steps, 1 if EMU else 2,
                           a.mixed_precision)
        except Exception as e:  # noqa: BLE001
            tr = {"error": f"{type(e).__name__}: {e}"}
        dog.cancel()
        if rank == 0:
            side = res.pop("_side_cpu", None) if isinstance(res, dict) else None
            if isinstance(side, dict) and "train" in side and isinstance(tr, dict):
                tr["cpu_baseline"] = side["train"]
            res["train"] = tr
    if rank == 0:
        res.pop("_side_cpu", None)
        print(json.dumps(res), flush=True)
    job.close()


if __name__ == "__main__":
    main()
