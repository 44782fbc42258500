#!/usr/bin/env python
"""Golden vectors produced by EXECUTING THE REFERENCE'S OWN TRAINING SCRIPT (tests/golden/reference_train.npz).

`/root/reference/scripts/train_unet.py` `main(args)` is imported from where it lies and run as written for two epochs of
three steps (five 32x32 spectrograms, batch 2 -> the last batch of an epoch is partial) on a dataset written by the reference's
own `scripts/audio_to_images.py`.  accelerate and datasets are the real packages; diffusers, librosa and torchvision are the
stand-ins of tests/refshim (UNet2DModel / DDPMScheduler = the oracle's; get_scheduler, EMAModel, ToTensor/Normalize restated
from their documentation).  So the fixture pins the script's own program text: the hard-coded
113.67 M-parameter UNet2DModel config, ToTensor+Normalize([0.5],[0.5]) preprocessing, noise / timestep sampling, add_noise,
MSE against the noise, clip_grad_norm_(1.0) on synchronising steps, AdamW with the script's default hyper-parameters, the
cosine schedule with warm-up whose length is len(dataloader)*epochs//accumulation, EMAModel(model, inv_gamma, power,
max_value) stepped after every optimizer step, and the EMA weights being copied INTO the live model at every save epoch.

Recorded: per step the clean batch, the noise, the timesteps (taken at the add_noise call), the logged loss / lr / ema_decay;
after each epoch a summary of every parameter tensor of the saved model (mean, L2 norm, first three entries).
Initial weights are not stored (455 MB): they are `torch.manual_seed(SEED); UNet2DModel(**UNET_KW)` of the oracle, and the
fixture holds their SHA-256.   Run:  python tests/golden/make_reference_train_golden.py [output.npz]
"""
import argparse
import hashlib
import importlib.util
import os
import sys
import tempfile

import numpy as np
import torch

sys.dont_write_bytecode = True      # the reference tree is read-only input: no __pycache__ next to its sources

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
SEED = 1234
RES = 32
UNET_KW = dict(sample_size=(RES, RES), in_channels=1, out_channels=1, layers_per_block=2,           # train_unet.py:116-139
               block_out_channels=(128, 128, 256, 256, 512, 512),
               down_block_types=("DownBlock2D",) * 4 + ("AttnDownBlock2D", "DownBlock2D"),
               up_block_types=("UpBlock2D", "AttnUpBlock2D") + ("UpBlock2D",) * 4)
ARGS = dict(local_rank=-1, dataset_config_name=None, train_data_dir=None, overwrite_output_dir=False, cache_dir=None,
            train_batch_size=2, eval_batch_size=2, num_epochs=2, save_images_epochs=1000, save_model_epochs=1,
            gradient_accumulation_steps=1, learning_rate=1e-4, lr_scheduler="cosine", lr_warmup_steps=2, adam_beta1=0.95,
            adam_beta2=0.999, adam_weight_decay=1e-6, adam_epsilon=1e-08, use_ema=True, ema_inv_gamma=1.0, ema_power=3 / 4,
            ema_max_decay=0.9999, push_to_hub=False, use_auth_token=False, hub_token=None, hub_model_id=None,
            hub_private_repo=False, logging_dir="logs", mixed_precision="no", hop_length=128, sample_rate=22050, n_fft=512,
            from_pretrained=None, start_epoch=0, num_train_steps=1000, scheduler="ddpm", vae=None, encodings=None)
# everything above except the data / epoch sizes is the script's argparse default (train_unet.py:357-417), lr_warmup_steps
# shortened from 500 so that six steps cross the warm-up / cosine boundary


def training_wav(path):
    """5.4 slices of a chirp over a noise floor at 22050 Hz, 16-bit PCM."""
    import wave
    rs = np.random.RandomState(5)
    n = RES * 128 - 1
    t = np.arange(int(5.4 * n)) / 22050
    x = 0.4 * np.sin(2 * np.pi * 500 * t * (1 + 4 * t)) + 0.05 * rs.standard_normal(len(t))
    with wave.open(path, "wb") as w:
        w.setnchannels(1), w.setsampwidth(2), w.setframerate(22050)
        w.writeframes(np.round(np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())


def state_digest(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(sd[k].detach().numpy().tobytes())
    return h.hexdigest()


def summarize(named):
    """(names, [mean, l2, first three entries]) per tensor, float64."""
    names, rows = [], []
    for k, v in named:
        f = v.detach().double().flatten()
        head = torch.zeros(3, dtype=torch.float64)
        head[: min(3, f.numel())] = f[:3]
        names.append(k)
        rows.append([float(f.mean()), float(f.norm())] + head.tolist())
    return np.array(names), np.array(rows)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main(out_path):
    sys.path[:0] = [os.path.join(ROOT, "tests", "refshim"), REFERENCE, ROOT]
    import huggingface_hub
    for gone in ("HfFolder", "Repository", "whoami"):      # names the script imports for --push_to_hub; removed upstream since
        if not hasattr(huggingface_hub, gone):
            setattr(huggingface_hub, gone, None)
    import accelerate
    import diffusers
    from oracle import schedulers as osched
    builder = _load("reference_audio_to_images", os.path.join(REFERENCE, "scripts", "audio_to_images.py"))
    train = _load("reference_train_unet", os.path.join(REFERENCE, "scripts", "train_unet.py"))

    steps, logs, saved = [], [], []
    orig_add_noise = osched._SchedulerBase.add_noise

    def recording_add_noise(self, original_samples, noise, timesteps):
        steps.append((original_samples.clone(), noise.clone(), timesteps.clone()))
        return orig_add_noise(self, original_samples, noise, timesteps)

    def recording_log(self, values, step=None, log_kwargs={}):
        logs.append(dict(values))

    def recording_save(self, output_dir):
        saved.append(summarize(self.unet.named_parameters()))

    init = {}
    orig_unet_init = diffusers.UNet2DModel.__init__

    def recording_unet_init(self, **kw):
        orig_unet_init(self, **kw)
        init["kw"], init["digest"] = kw, state_digest(self.state_dict())

    osched._SchedulerBase.add_noise = recording_add_noise
    accelerate.Accelerator.log = recording_log
    diffusers.DiffusionPipeline.save_pretrained = recording_save
    diffusers.UNet2DModel.__init__ = recording_unet_init
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "wav"))
        training_wav(os.path.join(tmp, "wav", "train.wav"))
        builder.main(argparse.Namespace(input_dir=os.path.join(tmp, "wav"), output_dir=os.path.join(tmp, "data"), resolution=(RES, RES),
                                        hop_length=128, push_to_hub=None, sample_rate=22050, n_fft=512))
        torch.manual_seed(SEED)
        train.main(argparse.Namespace(dataset_name=os.path.join(tmp, "data"), output_dir=os.path.join(tmp, "model"), **ARGS))
    assert init["kw"] == UNET_KW, init["kw"]
    assert len(steps) == len(logs) == 6 and len(saved) == 2
    out["init_sha256"] = np.array(init["digest"])
    for i, ((clean, noise, t), lg) in enumerate(zip(steps, logs)):
        out[f"step{i}:clean"], out[f"step{i}:noise"], out[f"step{i}:timesteps"] = clean.numpy(), noise.numpy(), t.numpy()
        out[f"step{i}:log"] = np.array([lg["loss"], lg["lr"], lg["ema_decay"], lg["step"]], dtype=np.float64)
    for e, (names, rows) in enumerate(saved):
        out[f"epoch{e}:names"], out[f"epoch{e}:summary"] = names, rows
    np.savez_compressed(out_path, **out)
    for i in range(6):
        print(i, out[f"step{i}:clean"].shape, out[f"step{i}:timesteps"], out[f"step{i}:log"])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "reference_train.npz"))
