"""scripts/train_unet.py end to end on the emulator: resume from a (tiny) saved pipeline, train on a dataset written by
scripts/audio_to_images.py with gradient accumulation + EMA, save in the diffusers layout, reload and sample."""
import importlib.util
import os

import numpy as np
import pytest
import scipy.io.wavfile
import torch

from native_backend import select

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = dict(sample_size=(16, 16), in_channels=1, out_channels=1, layers_per_block=1, block_out_channels=(32, 64),
            down_block_types=("DownBlock2D", "AttnDownBlock2D"), up_block_types=("AttnUpBlock2D", "UpBlock2D"))
MEL = dict(x_res=16, y_res=16, hop_length=64, n_fft=256, n_iter=2, sample_rate=4000)


def _script(name):
    spec = importlib.util.spec_from_file_location("adm_" + name, os.path.join(ROOT, "audio-diffusion_amd", "scripts", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_dataset_builder_then_training_script_roundtrip(tmp_path):
    select("emu")
    from audiodiffusion import AudioDiffusionPipeline, DDPMScheduler, Mel, UNet2DModel
    # 1. a dataset in the reference's on-disk format, from audio
    rng = np.random.default_rng(0)
    os.makedirs(tmp_path / "wav")
    slice_size = MEL["x_res"] * MEL["hop_length"] - 1
    for k in range(2):
        y = (0.3 * rng.standard_normal(slice_size * 4 + 5)).astype(np.float32)
        scipy.io.wavfile.write(tmp_path / "wav" / f"{k}.wav", MEL["sample_rate"], y)
    a2i = _script("audio_to_images")
    a2i.main(a2i.parse_args(["--input_dir", str(tmp_path / "wav"), "--output_dir", str(tmp_path / "data"), "--resolution", "16",
                             "--hop_length", "64", "--sample_rate", "4000", "--n_fft", "256"]))
    # 2. a starting checkpoint (tiny architecture; the script's own default is the 113.67 M-parameter one)
    start = UNet2DModel(**TINY).init_random(3)
    AudioDiffusionPipeline(None, start, Mel(**MEL), DDPMScheduler()).save_pretrained(str(tmp_path / "start"))
    w0 = {k: v.clone() for k, v in start.state_dict().items()}
    # 3. train: 8 images, batch 2, accumulate 2, 2 epochs, EMA
    tr = _script("train_unet")
    tr.main(tr.parse_args(["--from_pretrained", str(tmp_path / "start"), "--dataset_name", str(tmp_path / "data"),
                           "--output_dir", str(tmp_path / "out"), "--train_batch_size", "2", "--num_epochs", "2",
                           "--gradient_accumulation_steps", "2", "--save_model_epochs", "1", "--lr_warmup_steps", "1",
                           "--learning_rate", "1e-3", "--hop_length", "64", "--sample_rate", "4000", "--n_fft", "256"]))
    # 4. reload and sample
    pipe = AudioDiffusionPipeline.from_pretrained(str(tmp_path / "out"))
    pipe.set_progress_bar_config(disable=True)
    w1 = pipe.unet.state_dict()
    assert set(w1) == set(w0)
    changed = sum(float((w1[k] - w0[k]).abs().max()) > 0 for k in w0)
    assert changed >= 0.9 * len(w0), f"only {changed} of {len(w0)} tensors moved"
    assert all(torch.isfinite(v).all() for v in w1.values())
    noise = torch.randn(1, 1, 16, 16, generator=torch.Generator().manual_seed(0))
    images, (sr, audios) = pipe(batch_size=1, steps=3, noise=noise, return_dict=False)
    assert images[0].size == (16, 16) and sr == 4000 and audios[0].shape[-1] == MEL["hop_length"] * (MEL["x_res"] - 1)


def _rank_main(rank, world, port, start_dir, out_dir, res_dir):
    import sys
    for q in (ROOT, os.path.join(ROOT, "audio-diffusion_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, q)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), ADM_EMU_THREADS="2")
    torch.set_num_threads(1)
    from native_backend import select as sel
    sel("emu")
    tr = _script("train_unet")
    model = tr.main(tr.parse_args(["--from_pretrained", start_dir, "--dataset_name", "synthetic", "--resolution", "16",
                                   "--synthetic_size", "8", "--output_dir", out_dir, "--train_batch_size", "2",
                                   "--num_epochs", "1", "--save_model_epochs", "1", "--lr_warmup_steps", "1",
                                   "--learning_rate", "1e-3", "--hop_length", "64", "--sample_rate", "4000", "--n_fft", "256",
                                   "--mixed_precision", "fp16"]))     # GradScaler path: loss scale 65536, un-scale + clip in one
                                                                      # pass, overflow flag all-reduced (this width has few 16-bit-eligible convs)
    np.save(os.path.join(res_dir, f"flat{rank}.npy"), model.flat.data.cpu().numpy())


def test_training_script_two_ranks_stay_in_lockstep(tmp_path):
    """One process per rank over gloo (on the GPU box: RCCL): rank-sharded batches, bucketed all-reduce queued from the
    reverse pass, identical replicas after every step and after the EMA hand-over at the checkpoint."""
    import torch.multiprocessing as mp
    select("emu")
    from audiodiffusion import AudioDiffusionPipeline, DDPMScheduler, Mel, UNet2DModel
    start = UNet2DModel(**TINY).init_random(3)
    AudioDiffusionPipeline(None, start, Mel(**MEL), DDPMScheduler()).save_pretrained(str(tmp_path / "start"))
    os.makedirs(tmp_path / "res")
    ctx = mp.get_context("spawn")
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, str(tmp_path / "start"), str(tmp_path / "out"), str(tmp_path / "res")))
             for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=400)
        assert p.exitcode == 0
    f0, f1 = np.load(tmp_path / "res" / "flat0.npy"), np.load(tmp_path / "res" / "flat1.npy")
    assert np.array_equal(f0, f1), "replicas diverged"
    assert np.isfinite(f0).all()
    saved = AudioDiffusionPipeline.from_pretrained(str(tmp_path / "out")).unet.state_dict()
    moved = max(float((saved[k] - v).abs().max()) for k, v in start.state_dict().items())
    assert 0 < moved < 0.1


def test_latent_training_with_frozen_vae(tmp_path):
    """`--vae` (train_unet.py:95-104,231-235): the UNet trains on 0.18215 * posterior samples of a frozen AutoencoderKL, the
    saved pipeline carries the VAE and samples through encode -> denoise -> decode."""
    select("emu")
    from audiodiffusion import AudioDiffusionPipeline, AutoencoderKL
    vae_cfg = dict(sample_size=(32, 32), in_channels=1, out_channels=1, latent_channels=1, layers_per_block=1,
                   block_out_channels=(32, 64), down_block_types=("DownEncoderBlock2D",) * 2,
                   up_block_types=("UpDecoderBlock2D",) * 2)
    vae = AutoencoderKL(**vae_cfg).init_random(5)
    vae.save_pretrained(str(tmp_path / "vae"))
    from audiodiffusion import DDPMScheduler, Mel, UNet2DModel
    start = UNet2DModel(**dict(TINY, sample_size=16)).init_random(3)
    AudioDiffusionPipeline(vae, start, Mel(**dict(MEL, x_res=32, y_res=32)), DDPMScheduler()).save_pretrained(str(tmp_path / "start"))
    tr = _script("train_unet")
    model = tr.main(tr.parse_args(["--from_pretrained", str(tmp_path / "start"), "--dataset_name", "synthetic", "--resolution", "32",
                                   "--synthetic_size", "4", "--output_dir", str(tmp_path / "out"), "--train_batch_size", "2",
                                   "--num_epochs", "1", "--save_model_epochs", "1", "--lr_warmup_steps", "1",
                                   "--learning_rate", "1e-3", "--hop_length", "64", "--sample_rate", "4000", "--n_fft", "256"]))
    assert tuple(model._hw()) == (16, 16)                      # trained at the latent resolution (32 / 2)
    out = AudioDiffusionPipeline.from_pretrained(str(tmp_path / "out"))
    assert out.vqvae is not None and tuple(out.unet._hw()) == (16, 16)
    out.set_progress_bar_config(disable=True)
    images, _ = out(batch_size=1, steps=2, generator=torch.Generator().manual_seed(0), audio=False, return_float=True)
    assert images[0].size == (32, 32)
    moved = max(float((out.unet.state_dict()[k] - v).abs().max()) for k, v in start.state_dict().items())
    assert 0 < moved < 0.1


def test_conditional_training_from_encodings(tmp_path):
    """encode_audio.py -> pickled {audio_file: (1, 100) encoding} -> train_unet.py --encodings (scripts/encode_audio.py:27-30,
    scripts/train_unet.py:85-87,93-94,254-255) -> saved conditional pipeline samples with `encoding=`."""
    import pickle
    select("emu")
    from audiodiffusion import AudioDiffusionPipeline, AudioEncoder, DDPMScheduler, Mel, UNet2DConditionModel
    from oracle import audio_encoder as oenc
    rng = np.random.default_rng(0)
    os.makedirs(tmp_path / "wav")
    slice_size = MEL["x_res"] * MEL["hop_length"] - 1
    for k in range(2):
        y = (0.3 * rng.standard_normal(slice_size * 3 + 5)).astype(np.float32)
        scipy.io.wavfile.write(tmp_path / "wav" / f"{k}.wav", MEL["sample_rate"], y)
    a2i = _script("audio_to_images")
    a2i.main(a2i.parse_args(["--input_dir", str(tmp_path / "wav"), "--output_dir", str(tmp_path / "data"), "--resolution", "16",
                             "--hop_length", "64", "--sample_rate", "4000", "--n_fft", "256"]))
    # encodings: the reference's AudioEncoder architecture on a small mel geometry
    enc = AudioEncoder()
    enc.mel = Mel(x_res=24, y_res=16, sample_rate=4000, n_fft=256, hop_length=64, top_db=80)
    enc.load_state_dict(oenc.random_state_dict(3, y_res=16, x_res=24))
    ea = _script("encode_audio")
    table = ea.main(ea.parse_args(["--dataset_name", str(tmp_path / "data"), "--output_file", str(tmp_path / "enc.p")]), enc)
    assert len(table) == 2 and all(tuple(v.shape) == (1, 100) for v in table.values())
    with open(tmp_path / "enc.p", "rb") as f:
        assert set(pickle.load(f)) == set(table)
    cfg = dict(sample_size=(16, 16), in_channels=1, out_channels=1, layers_per_block=1, block_out_channels=(32, 64),
               down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"),
               cross_attention_dim=100, attention_head_dim=4)
    start = UNet2DConditionModel(**cfg).init_random(3)
    AudioDiffusionPipeline(None, start, Mel(**MEL), DDPMScheduler()).save_pretrained(str(tmp_path / "start"))
    tr = _script("train_unet")
    tr.main(tr.parse_args(["--from_pretrained", str(tmp_path / "start"), "--dataset_name", str(tmp_path / "data"),
                           "--encodings", str(tmp_path / "enc.p"), "--output_dir", str(tmp_path / "out"),
                           "--train_batch_size", "2", "--num_epochs", "1", "--save_model_epochs", "1", "--lr_warmup_steps", "1",
                           "--learning_rate", "1e-3", "--hop_length", "64", "--sample_rate", "4000", "--n_fft", "256"]))
    pipe = AudioDiffusionPipeline.from_pretrained(str(tmp_path / "out"))
    assert type(pipe.unet).__name__ == "UNet2DConditionModel"
    moved = sum(float((pipe.unet.state_dict()[k] - v).abs().max()) > 0 for k, v in start.state_dict().items())
    assert moved >= 0.85 * len(start.state_dict())      # to_k of a 1-token encoding has no gradient, everything else moves
    pipe.set_progress_bar_config(disable=True)
    e = next(iter(table.values()))[:, None, :]
    images, _ = pipe(batch_size=1, steps=2, generator=torch.Generator().manual_seed(0), encoding=e, audio=False,
                     return_float=True)
    assert images[0].size == (16, 16)


def test_sample_files_at_save_images_epochs(tmp_path):
    """`--save_images_epochs` (train_unet.py:313-348): samples from the live (EMA) weights with the fixed seed, here written as
    PNG / WAV files instead of tensorboard events; the training model itself runs the sampling loop."""
    select("emu")
    from audiodiffusion import AudioDiffusionPipeline, DDIMScheduler, Mel, UNet2DModel
    start = UNet2DModel(**TINY).init_random(3)
    AudioDiffusionPipeline(None, start, Mel(**MEL), DDIMScheduler()).save_pretrained(str(tmp_path / "start"))
    tr = _script("train_unet")
    tr.main(tr.parse_args(["--from_pretrained", str(tmp_path / "start"), "--dataset_name", "synthetic", "--resolution", "16",
                           "--synthetic_size", "4", "--output_dir", str(tmp_path / "out"), "--train_batch_size", "2",
                           "--num_epochs", "1", "--save_model_epochs", "5", "--save_images_epochs", "1", "--eval_batch_size", "2",
                           "--scheduler", "ddim", "--lr_warmup_steps", "1", "--hop_length", "64", "--sample_rate", "4000",
                           "--n_fft", "256"]))
    files = sorted(os.listdir(tmp_path / "out" / "samples"))
    assert files == ["epoch0000_0.png", "epoch0000_0.wav", "epoch0000_1.png", "epoch0000_1.wav"]
    sr, audio = scipy.io.wavfile.read(tmp_path / "out" / "samples" / "epoch0000_1.wav")
    assert sr == 4000 and audio.dtype == np.float32 and abs(float(np.abs(audio).max()) - 1.0) < 1e-6
    from PIL import Image
    assert Image.open(tmp_path / "out" / "samples" / "epoch0000_0.png").size == (16, 16)
    assert os.path.exists(tmp_path / "out" / "unet" / "config.json")            # last epoch also saves the model
