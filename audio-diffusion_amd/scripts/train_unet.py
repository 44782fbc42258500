#!/usr/bin/env python
"""MI355X-native drop-in for `scripts/train_unet.py` of teticio/audio-diffusion (unconditional pixel-space path).

Same CLI flag names and defaults as the reference (`scripts/train_unet.py:354-420`) for everything on the hot path; the
training step (`:226-267`) runs as: fused add_noise kernel -> native UNet forward+backward (`adm_unet_forward_backward`)
-> bucketed gradient all-reduce over RCCL/xGMI (`torch.distributed`, one process per GPU; replaces accelerate's DDP,
`config/accelerate_multi_gpu.yaml`) -> fused clip + AdamW + EMA kernel -> weight re-pack. Checkpoints are written in the
diffusers layout by `AudioDiffusionPipeline.save_pretrained` every `--save_model_epochs` (`:286-311`).

Launch:  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train_unet.py --dataset_name ...
Implemented: fp32 and --mixed_precision bf16, --vae (latent training), --encodings (conditional training), gradient
accumulation, EMA, DDP. Not implemented (raise): fp16; hub push and tensorboard are ignored.
"""
import argparse
import pickle
import math
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from audiodiffusion import (AudioDiffusionPipeline, AutoencoderKL, DDIMScheduler, DDPMScheduler, Mel,  # noqa: E402
                            UNet2DConditionModel, UNet2DModel)
from audiodiffusion import training as T  # noqa: E402


def synthetic_dataset(n, resolution, seed=7):
    """N u8 images ~ round(255*sigmoid(N(0,1) smoothed 5x5)), fixed seed."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 1, resolution[0], resolution[1], generator=g)
    k = torch.ones(1, 1, 5, 5) / 25.0
    x = torch.nn.functional.conv2d(x, k, padding=2) * 5.0
    return torch.round(255 * torch.sigmoid(x)).to(torch.uint8)


def load_images(args, resolution):
    """-> ((N,1,H,W) uint8 images, audio_file per image) — the two dataset columns the reference's transform reads (:80-87)."""
    if args.dataset_name == "synthetic":
        return synthetic_dataset(args.synthetic_size, resolution), [f"synthetic_{i % 4}" for i in range(args.synthetic_size)]
    from datasets import load_dataset, load_from_disk
    if os.path.exists(args.dataset_name):
        ds = load_from_disk(args.dataset_name)["train"]
    else:
        ds = load_dataset(args.dataset_name, args.dataset_config_name, cache_dir=args.cache_dir, split="train")
    imgs = [np.frombuffer(im.tobytes(), dtype="uint8").reshape(im.height, im.width) for im in ds["image"]]
    files = list(ds["audio_file"]) if "audio_file" in ds.column_names else [None] * len(imgs)
    return torch.from_numpy(np.stack(imgs))[:, None], files


class Trainer:
    """One rank's training state and the step of the reference's loop body (train_unet.py:226-267) on given tensors:
    add_noise -> native forward + backward -> (accumulate) -> bucketed all-reduce -> clip_grad_norm_(1.0) -> AdamW -> LR
    schedule -> EMA -> weight re-pack.  `main` feeds it batches; tests/test_reference_train_pin.py feeds it the batches, noise
    and timesteps recorded from a run of the reference's own script."""

    def __init__(self, args, model, resolution, steps_per_epoch, world=1, dev=None):
        self.args, self.model, self.world, self.dev = args, model, world, dev
        # identical init on every rank (same seed / checkpoint). bf16: eligible 3x3 convolutions (forward, data and weight
        # gradient) run on bf16 MFMA operands with fp32 accumulation; master weights, optimizer state and gradients stay fp32
        self.flat, self.grads = model.enable_training(resolution, mixed_precision=args.mixed_precision)
        self.optimizer = T.AdamW(self.flat, lr=args.learning_rate, betas=(args.adam_beta1, args.adam_beta2),
                                 weight_decay=args.adam_weight_decay, eps=args.adam_epsilon)
        # the LR scheduler steps on synchronising steps only (:178): its length is divided by the accumulation factor
        num_training_steps = (steps_per_epoch * args.num_epochs) // args.gradient_accumulation_steps
        self.lr_scheduler = T.LambdaLR(self.optimizer, T.get_cosine_schedule_with_warmup(args.lr_warmup_steps, num_training_steps)
                                       if args.lr_scheduler == "cosine" else (lambda s: 1.0))
        self.ema = T.EMAModel(self.flat, inv_gamma=args.ema_inv_gamma, power=args.ema_power,
                              max_value=args.ema_max_decay) if args.use_ema else None
        self.reducer = T.GradAllReducer(self.grads)
        if world > 1 and args.gradient_accumulation_steps == 1:
            self.reducer.attach(model)   # DDP overlap: buckets are all-reduced while the reverse pass is still running
        self.accum = T.GradAccumulator(self.grads, args.gradient_accumulation_steps)   # accelerator.accumulate(model), :252
        self.scaler = T.GradScaler() if args.mixed_precision == "fp16" else None       # accelerate's GradScaler under fp16 (:391-395)

    def step(self, noise_scheduler, clean, noise, timesteps, encoding=None, last_batch=False):
        model, ema, grads, flat, scaler = self.model, self.ema, self.grads, self.flat, self.scaler
        noisy = noise_scheduler.add_noise(clean.contiguous(), noise, timesteps)
        self.reducer.begin_step()
        ls = scaler.get_scale() if scaler is not None else 1.0
        if encoding is not None:                                       # :254-255
            loss = model.train_step(noisy, timesteps, noise, encoding, loss_scale=ls)
        else:
            loss = model.train_step(noisy, timesteps, noise, loss_scale=ls)
        if self.accum.add(last_batch=last_batch):                      # accelerator.sync_gradients
            self.reducer.start()
            self.reducer.finish()
            found_inf = False
            if scaler is not None:                                     # unscale_ + clip_grad_norm_ in one pass; overflow check
                clip, found_inf = scaler.unscale_and_clip_(grads, 1.0)
                if self.world > 1:                                     # every rank takes the same decision
                    flag = torch.tensor([float(found_inf)], device=self.dev)
                    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                    found_inf = bool(flag.item())
                scaler.update(found_inf)
            else:
                clip = T.clip_grad_norm_(grads, 1.0)
            if found_inf:                                              # GradScaler.step skips the optimizer; accelerate then
                if ema is not None:                                    # skips the LR scheduler too; the EMA still steps
                    ema.step(flat)
            else:
                self.optimizer.step(grads, clip=clip, ema=ema, ema_decay=ema.next_decay() if ema is not None else 0.0)
                self.lr_scheduler.step()
                model.refresh_weights()
        elif ema is not None:                                          # micro-step: no collective (no_sync), optimizer and
            ema.step(flat)                                             # scheduler skipped, EMA still stepped (:265-266)
        return loss

    def copy_ema_into_model(self):
        """:292-294: at every save epoch the EMA weights go INTO the live model, and training continues from them."""
        if self.ema is not None:
            self.ema.copy_to(self.flat)
            self.model.refresh_weights()
        self.model.sync_state_dict_from_flat()


def main(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    on_gpu = torch.cuda.is_available()
    if on_gpu:
        torch.cuda.set_device(local)
    dev = torch.device("cuda", local) if on_gpu else torch.device("cpu")
    if world > 1:
        dist.init_process_group("nccl" if on_gpu else "gloo")
    output_dir = os.environ.get("SM_MODEL_DIR", None) or args.output_dir

    resolution = (args.resolution, args.resolution) if isinstance(args.resolution, int) else args.resolution
    images, audio_files = load_images(args, resolution)            # (N,1,H,W) uint8
    resolution = tuple(images.shape[2:])
    # conditional training (:93-94): pickled {audio_file: encoding} as written by scripts/encode_audio.py of the reference
    enc_table = None
    if args.encodings is not None:
        with open(args.encodings, "rb") as f:
            encodings = pickle.load(f)
        missing = sorted({a for a in audio_files if a not in encodings})
        if missing:
            raise KeyError(f"--encodings has no entry for {missing[:3]}{'...' if len(missing) > 3 else ''}")
        enc_table = torch.stack([torch.as_tensor(np.asarray(encodings[a]), dtype=torch.float32).reshape(-1, np.asarray(
            encodings[a]).shape[-1]) for a in audio_files])           # (N, seq_length, cross_attention_dim)
    # latent diffusion (train_unet.py:95-104): a frozen AutoencoderKL maps every batch to latents the UNet is trained on
    vqvae, latent_resolution = None, None
    if args.vae is not None:
        try:
            vqvae = AutoencoderKL.from_pretrained(args.vae)
        except EnvironmentError:
            vqvae = AudioDiffusionPipeline.from_pretrained(args.vae).vqvae
        latent_resolution = tuple(vqvae.encode(torch.zeros((1, 1) + resolution, device=dev)).latent_dist.sample().shape[2:])
    if args.from_pretrained is not None:
        pipeline = AudioDiffusionPipeline.from_pretrained(args.from_pretrained)
        mel, model = pipeline.mel, pipeline.unet
        if getattr(pipeline, "vqvae", None) is not None:            # :110-111
            vqvae = pipeline.vqvae
            latent_resolution = tuple(vqvae.encode(torch.zeros((1, 1) + resolution, device=dev)).latent_dist.sample().shape[2:])
    else:
        lc = 1 if vqvae is None else vqvae.config["latent_channels"]
    if args.from_pretrained is None and enc_table is not None:       # :139-159
        model = UNet2DConditionModel(sample_size=resolution if vqvae is None else latent_resolution, in_channels=lc,
                                     out_channels=lc, layers_per_block=2, block_out_channels=(128, 256, 512, 512),
                                     down_block_types=("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",),
                                     up_block_types=("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3,
                                     cross_attention_dim=int(enc_table.shape[-1])).init_random(args.seed)
        mel = Mel(x_res=resolution[1], y_res=resolution[0], hop_length=args.hop_length, sample_rate=args.sample_rate,
                  n_fft=args.n_fft)
    elif args.from_pretrained is None:
        model = UNet2DModel(sample_size=resolution if vqvae is None else latent_resolution, in_channels=lc,
                            out_channels=lc, layers_per_block=2,
                            block_out_channels=(128, 128, 256, 256, 512, 512),
                            down_block_types=("DownBlock2D",) * 4 + ("AttnDownBlock2D", "DownBlock2D"),
                            up_block_types=("UpBlock2D", "AttnUpBlock2D") + ("UpBlock2D",) * 4).init_random(args.seed)
        mel = Mel(x_res=resolution[1], y_res=resolution[0], hop_length=args.hop_length, sample_rate=args.sample_rate,
                  n_fft=args.n_fft)
    noise_scheduler = (DDPMScheduler if args.scheduler == "ddpm" else DDIMScheduler)(num_train_timesteps=args.num_train_steps)

    n_local = len(images) // world
    # len(train_dataloader) of the reference (:91, DataLoader default drop_last=False): the last batch may be partial
    steps_per_epoch = -(-n_local // args.train_batch_size)
    if steps_per_epoch == 0:
        raise ValueError(f"dataset of {len(images)} images leaves rank {rank} of {world} without a single sample")
    tr = Trainer(args, model, resolution if vqvae is None else latent_resolution, steps_per_epoch, world, dev)
    flat, ema, lr_scheduler = tr.flat, tr.ema, tr.lr_scheduler

    global_step = 0
    for epoch in range(args.num_epochs):
        if epoch < args.start_epoch:                               # train_unet.py:216-224: replay the LR schedule
            for _ in range(steps_per_epoch):
                lr_scheduler.step()
                global_step += 1
            if ema is not None:
                ema.optimization_step = global_step
            continue
        g = torch.Generator().manual_seed(args.seed + epoch)
        perm = torch.randperm(len(images), generator=g)[rank::world][:n_local]   # DataLoader(shuffle=True), sharded by rank
        t0, seen = time.perf_counter(), 0
        for it in range(steps_per_epoch):
            idx = perm[it * args.train_batch_size:(it + 1) * args.train_batch_size]
            clean = (images[idx].to(dev).float() / 255.0 - 0.5) / 0.5     # ToTensor + Normalize([0.5],[0.5]) (:73-78)
            if vqvae is not None:        # :231-235: posterior sample, scaled to roughly unit variance
                clean = vqvae.encode(clean.contiguous()).latent_dist.sample() * 0.18215
            noise = torch.randn(clean.shape).to(dev)                          # CPU RNG then H2D, as :238
            timesteps = torch.randint(0, noise_scheduler.config.num_train_timesteps, (clean.shape[0],)).long()
            loss = tr.step(noise_scheduler, clean, noise, timesteps, None if enc_table is None else enc_table[idx].to(dev),
                           last_batch=(it == steps_per_epoch - 1))
            global_step += 1
            seen += clean.shape[0] * world
            if rank == 0 and (it % args.log_every == 0 or it == steps_per_epoch - 1):
                print(f"epoch {epoch} step {global_step} loss {float(loss):.5f} lr {lr_scheduler.get_last_lr()[0]:.3e} "
                      f"ema_decay {ema.cur_decay_value if ema else 0:.5f} {seen / (time.perf_counter() - t0):.1f} samples/s",
                      flush=True)
        if world > 1:
            dist.barrier()
        save_model = (epoch + 1) % args.save_model_epochs == 0 or epoch == args.num_epochs - 1
        save_images = (epoch + 1) % args.save_images_epochs == 0
        if rank == 0 and (save_model or save_images):               # :286-298
            tr.copy_ema_into_model()                                # :292-294
            pipeline = AudioDiffusionPipeline(vqvae=vqvae, unet=model, mel=mel, scheduler=noise_scheduler)
            if save_model:
                pipeline.save_pretrained(output_dir)
            if save_images:                                         # :313-348 (the reference logs these to tensorboard)
                write_samples(pipeline, args, epoch, output_dir, dev,
                              None if enc_table is None else [enc_table[i] for i in range(len(enc_table))])
        if world > 1:
            dist.broadcast(flat, src=0)                             # keep replicas identical after the EMA copy
            model.refresh_weights()
            dist.barrier()
    if world > 1:
        dist.destroy_process_group()
    return model


def write_samples(pipeline, args, epoch, output_dir, dev, encodings):
    """`eval_batch_size` samples from the current (EMA) weights with the reference's fixed seeds (generator 42,
    random.seed(42) for the encodings, :313-329), written as PNG + peak-normalised WAV files under
    <output_dir>/samples/ — the stand-in for the tensorboard images / audio of :332-347 (no tensorboard here)."""
    import random
    import scipy.io.wavfile
    generator = torch.Generator(device=dev).manual_seed(42)            # on the training device, as :314
    encoding = None
    if encodings is not None:
        random.seed(42)
        encoding = torch.stack(random.sample(encodings, min(args.eval_batch_size, len(encodings)))).to(dev)
    n = args.eval_batch_size if encoding is None else encoding.shape[0]
    pipeline.set_progress_bar_config(disable=True)
    images, (sample_rate, audios) = pipeline(generator=generator, batch_size=n, return_dict=False, encoding=encoding)
    d = os.path.join(output_dir, "samples")
    os.makedirs(d, exist_ok=True)
    for i, (image, audio) in enumerate(zip(images, audios)):
        image.save(os.path.join(d, f"epoch{epoch:04d}_{i}.png"))
        peak = float(np.abs(audio).max())
        scipy.io.wavfile.write(os.path.join(d, f"epoch{epoch:04d}_{i}.wav"), sample_rate,
                               (audio / peak if peak > 0 else audio).astype(np.float32))       # librosa.util.normalize


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="MI355X-native training script (reference: scripts/train_unet.py).")
    parser.add_argument("--local_rank", type=int, default=-1)
    parser.add_argument("--dataset_name", type=str, default="synthetic")
    parser.add_argument("--dataset_config_name", type=str, default=None)
    parser.add_argument("--output_dir", type=str, default="ddpm-model-64")
    parser.add_argument("--overwrite_output_dir", type=bool, default=False)
    parser.add_argument("--cache_dir", type=str, default=None)
    parser.add_argument("--train_batch_size", type=int, default=16)
    parser.add_argument("--eval_batch_size", type=int, default=16)
    parser.add_argument("--num_epochs", type=int, default=100)
    parser.add_argument("--save_images_epochs", type=int, default=10)
    parser.add_argument("--save_model_epochs", type=int, default=10)
    parser.add_argument("--gradient_accumulation_steps", type=int, default=1)
    parser.add_argument("--learning_rate", type=float, default=1e-4)
    parser.add_argument("--lr_scheduler", type=str, default="cosine")
    parser.add_argument("--lr_warmup_steps", type=int, default=500)
    parser.add_argument("--adam_beta1", type=float, default=0.95)
    parser.add_argument("--adam_beta2", type=float, default=0.999)
    parser.add_argument("--adam_weight_decay", type=float, default=1e-6)
    parser.add_argument("--adam_epsilon", type=float, default=1e-08)
    parser.add_argument("--use_ema", type=bool, default=True)
    parser.add_argument("--ema_inv_gamma", type=float, default=1.0)
    parser.add_argument("--ema_power", type=float, default=3 / 4)
    parser.add_argument("--ema_max_decay", type=float, default=0.9999)
    parser.add_argument("--mixed_precision", type=str, default="no", choices=["no", "fp16", "bf16"])
    parser.add_argument("--hop_length", type=int, default=512)
    parser.add_argument("--sample_rate", type=int, default=22050)
    parser.add_argument("--n_fft", type=int, default=2048)
    parser.add_argument("--from_pretrained", type=str, default=None)
    parser.add_argument("--start_epoch", type=int, default=0)
    parser.add_argument("--num_train_steps", type=int, default=1000)
    parser.add_argument("--scheduler", type=str, default="ddpm", help="ddpm or ddim")
    parser.add_argument("--vae", type=str, default=None)
    parser.add_argument("--encodings", type=str, default=None)
    # additions (not in the reference)
    parser.add_argument("--resolution", type=int, default=256, help="image size of the synthetic dataset")
    parser.add_argument("--synthetic_size", type=int, default=2048)
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--log_every", type=int, default=10)
    return parser.parse_args(argv)


if __name__ == "__main__":
    main(parse_args())
