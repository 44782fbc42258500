#!/usr/bin/env python
"""Golden vectors produced by EXECUTING THE REFERENCE'S OWN CODE (tests/golden/reference_pipeline.npz).

`/root/reference/audiodiffusion/pipeline_audio_diffusion.py` (AudioDiffusionPipeline.__call__ / encode / slerp) and
`/root/reference/audiodiffusion/mel.py` (Mel) are imported from where they lie and run as written.  Their two third-party
imports, diffusers 0.24 and librosa 0.10.2, are not in this image; `tests/refshim/` supplies stand-ins whose classes and
functions forward to the CPU oracle.  So what this fixture pins is the reference's own program text on the path — argument
handling, noise aliasing (`images = noise`, the in-place `images[0, 0] = ...`), the audio-conditioned start, the mask
tensor and the order it is applied in, latent scaling by 0.18215, `(x*255).round()` to uint8, the DDIM inversion loop,
slerp, and the Mel class's slicing / zero padding / byte <-> dB conversions — while the UNet, scheduler, VAE, STFT/NNLS/
Griffin-Lim arithmetic underneath is the oracle's (unpinned, except the Mel forward rows: make_thirdparty_mel.py).

tests/test_reference_pin.py then requires (a) the oracle's restatement of the pipeline to reproduce these numbers exactly,
(b) the product path (HIP kernels through the C-ABI) to reproduce them within the path's tolerances, and (c) — wherever
/root/reference is present — this script to regenerate the committed file bit for bit.
Run:  python tests/golden/make_reference_golden.py [output.npz]
"""
import os
import sys
import tempfile

import numpy as np
import torch
from PIL import Image

sys.dont_write_bytecode = True      # the reference tree is read-only input: no __pycache__ next to its sources

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"

UNET_CFG = dict(sample_size=16, in_channels=1, out_channels=1, layers_per_block=1, block_out_channels=(32, 32),
                down_block_types=("DownBlock2D", "AttnDownBlock2D"), up_block_types=("AttnUpBlock2D", "UpBlock2D"))
MEL16 = dict(x_res=16, y_res=16, hop_length=128, n_fft=512, n_iter=4, sample_rate=22050)
VAE_CFG = dict(sample_size=(32, 32), in_channels=1, out_channels=1, latent_channels=1, layers_per_block=1,
               block_out_channels=(32, 32), down_block_types=("DownEncoderBlock2D",) * 2,
               up_block_types=("UpDecoderBlock2D",) * 2)
MEL32 = dict(x_res=32, y_res=32, hop_length=128, n_fft=512, n_iter=4, sample_rate=22050)


def pcm(v):
    return (np.round(v * 32767.0).clip(-32768, 32767) / 32768.0).astype(np.float32)


def phases(rng, mel_cfg, n):
    return rng.random_sample((n, mel_cfg["n_fft"] // 2 + 1, mel_cfg["x_res"])).astype(np.float32).astype(np.float64)


def u8(images):
    return np.stack([np.asarray(i) for i in images])


DATASET_RES = (32, 16)
COND_SEED = 21
COND_CFG = dict(sample_size=16, in_channels=1, out_channels=1, layers_per_block=1, block_out_channels=(32, 64),
                down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"),
                cross_attention_dim=12, attention_head_dim=4)


# ---- the notebook cells (case N)
MEL_NB = dict(x_res=32, y_res=16, hop_length=1024, n_fft=2048, n_iter=4, sample_rate=8000)   # 4.1 s slices: the cells' 2 s overlap fits
NB_SEED, NB_START_STEP, NB_REMIX_SLICES, NB_CALLS, NB_REMIX_SEED = 31, 3, 4, 18, 20260924


def notebook_phases():
    """Griffin-Lim start phases of the NB_CALLS pipeline calls of case N (frozen RandomState stream; float32-exact values)."""
    return np.random.RandomState(41).random_sample((NB_CALLS, MEL_NB["n_fft"] // 2 + 1, MEL_NB["x_res"])).astype(np.float32)


def notebook_cells():
    import json
    with open(os.path.join(REFERENCE, "notebooks", "audio_diffusion_pipeline.ipynb")) as f:
        return ["".join(c["source"]) for c in json.load(f)["cells"] if c["cell_type"] == "code"]


def cell_source(cells, marker, params):
    """The one code cell containing `marker`, as written; a `name = value  #@param ...` form line takes params[name] if given."""
    hits = [c for c in cells if marker in c]
    assert len(hits) == 1, (marker, len(hits))
    lines = []
    for line in hits[0].split("\n"):
        name = line.split("=")[0].strip()
        if "#@param" in line and name in params:
            line = f"{name} = {params[name]!r}"
        lines.append(line)
    return "\n".join(lines)


def notebook_clip(n_slices, seed):
    rs = np.random.RandomState(seed)
    n = int(n_slices * MEL_NB["x_res"] * MEL_NB["hop_length"])
    t = np.arange(n) / MEL_NB["sample_rate"]
    return pcm(0.3 * np.sin(2 * np.pi * 300 * t * (1 + 0.3 * t)) + 0.1 * np.sin(2 * np.pi * 1234 * t) + 0.03 * rs.standard_normal(n))


def notebook_wav(path):
    import wave
    x = notebook_clip(2.3, 4)
    with wave.open(path, "wb") as w:
        w.setnchannels(1), w.setsampwidth(2), w.setframerate(MEL_NB["sample_rate"])
        w.writeframes(np.round(x * 32768.0).astype("<i2").tobytes())
    return x


def notebook_images():
    rs = np.random.RandomState(6)
    base = rs.randint(0, 256, (2, MEL_NB["y_res"], MEL_NB["x_res"])).astype(np.float64)
    ramp = np.linspace(0, 120, MEL_NB["x_res"])[None, None, :]
    return np.clip(0.5 * base + ramp, 0, 255).astype(np.uint8)


def encoder_wav(path):
    """2.4 slices of the encoder's Mel (216 frames x hop 512 at 22050 Hz), 16-bit PCM."""
    import wave
    rs = np.random.RandomState(12)
    n = int(2.4 * 216 * 512)
    t = np.arange(n) / 22050
    x = 0.3 * np.sin(2 * np.pi * 220 * t * (1 + 0.2 * t)) + 0.2 * np.sin(2 * np.pi * 2500 * t) * (t % 0.5 < 0.1) + 0.02 * rs.standard_normal(n)
    with wave.open(path, "wb") as w:
        w.setnchannels(1), w.setsampwidth(2), w.setframerate(22050)
        w.writeframes(np.round(np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())


def state_sha256(sd):
    import hashlib
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(sd[k].detach().numpy().tobytes())
    return h.hexdigest()


def cond_unet():
    from oracle.unet_condition import UNet2DConditionModel
    torch.manual_seed(COND_SEED)
    m = UNet2DConditionModel(**COND_CFG).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "norm" in n:
                p.add_(0.2 * torch.randn_like(p))
    return m


def dataset_wavs(root):
    """The input directory of case I (also rebuilt by the tests): {relative path: sha256 of the file}."""
    import hashlib
    import wave
    rs = np.random.RandomState(99)
    n = DATASET_RES[0] * 128 - 1                                  # slice_size at hop 128
    t = np.arange(int(2.6 * n)) / 22050
    a = 0.4 * np.sin(2 * np.pi * 800 * t * (1 + 2 * t)) + 0.02 * rs.standard_normal(len(t))
    b = np.concatenate([np.zeros(n), 0.1 * rs.standard_normal(n + 40)])
    files = {"a.wav": a, os.path.join("sub", "b.WAV"): b}
    digests = {}
    os.makedirs(os.path.join(root, "sub"), exist_ok=True)
    for rel, x in files.items():
        with wave.open(os.path.join(root, rel), "wb") as w:
            w.setnchannels(1), w.setsampwidth(2), w.setframerate(22050)
            w.writeframes(np.round(np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())
    with open(os.path.join(root, "c.mp3"), "wb") as f:
        f.write(b"not an audio file")
    for rel in list(files) + ["c.mp3"]:
        with open(os.path.join(root, rel), "rb") as f:
            digests[rel] = hashlib.sha256(f.read()).hexdigest()
    return digests


def main(out_path):
    sys.path[:0] = [os.path.join(ROOT, "tests", "refshim"), REFERENCE, ROOT]
    import audiodiffusion as ref                       # the reference package itself
    assert os.path.realpath(ref.__file__).startswith(REFERENCE + "/"), ref.__file__
    from librosa.feature import inverse as gl          # the stand-in: queue of Griffin-Lim start phases
    from oracle.schedulers import DDIMScheduler, DDPMScheduler
    from oracle.unet import UNet2DModel
    from oracle.vae import AutoencoderKL
    RefPipeline, RefMel = ref.AudioDiffusionPipeline, ref.mel.Mel

    z = np.load(os.path.join(HERE, "unet_tiny.npz"))
    unet = UNet2DModel(**UNET_CFG).eval()
    unet.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")})
    rs = np.random.RandomState(77)
    out = {}

    def fresh_unet():
        unet.sample_size = UNET_CFG["sample_size"]     # __call__ rewrites it to a tuple (:118-119)
        return unet

    # ---- A: DDIM, given noise, eta = 0; images + audio (return_dict=False) and the dict form
    g = torch.Generator().manual_seed(42)
    noise = torch.randn(2, 1, 16, 16, generator=g)
    ph = phases(rs, MEL16, 2)
    pipe = RefPipeline(None, fresh_unet(), RefMel(**MEL16), DDIMScheduler())
    gl.INIT_PHASES[:] = list(ph)
    images, (sr, audios) = pipe(batch_size=2, steps=4, noise=noise.clone(), return_dict=False)
    out.update({"A:noise": noise.numpy(), "A:phase": ph.astype(np.float32), "A:images": u8(images), "A:audios": np.stack(audios), "A:sr": np.array(sr)})
    gl.INIT_PHASES[:] = list(ph)
    d = pipe(batch_size=2, steps=4, noise=noise.clone())
    assert np.array_equal(u8(d.images), out["A:images"]) and d.audios.shape == (2, 1, out["A:audios"].shape[1])
    out["A:default_steps"] = np.array(pipe.get_default_steps())

    # ---- B: DDIM eta = 0.7, initial noise and per-step noise drawn from one generator
    pipe = RefPipeline(None, fresh_unet(), RefMel(**MEL16), DDIMScheduler())
    gl.INIT_PHASES[:] = list(ph)
    images, _ = pipe(batch_size=2, steps=4, eta=0.7, generator=torch.Generator().manual_seed(7), return_dict=False)
    out["B:images"] = u8(images)

    # ---- C: DDPM, 3 steps, separate generators for the start and for the steps
    pipe = RefPipeline(None, fresh_unet(), RefMel(**MEL16), DDPMScheduler())
    gl.INIT_PHASES[:] = list(ph)
    images, _ = pipe(batch_size=2, steps=3, generator=torch.Generator().manual_seed(11),
                     step_generator=torch.Generator().manual_seed(12), return_dict=False)
    out["C:images"] = u8(images)
    out["C:default_steps"] = np.array(pipe.get_default_steps())

    # ---- D: audio-conditioned start (slice 1 of a 2.3-slice clip), start_step 2 of 6, both masks.  Batch 1: the
    # reference's in-place `images[0, 0] = add_noise(...)` (:150) and its mask writes (:182-185) only broadcast for one row
    # (D3 / D4 record that both raise for two); D2 is the form without a start step (masks only)
    n_s = MEL16["x_res"] * MEL16["hop_length"]
    t = np.arange(int(2.3 * n_s)) / MEL16["sample_rate"]
    raw = pcm(0.3 * np.sin(2 * np.pi * 1200 * t * (1 + 3 * t)) + 0.05 * rs.standard_normal(len(t)))
    g = torch.Generator().manual_seed(3)
    noise = torch.randn(2, 1, 16, 16, generator=g)
    pipe = RefPipeline(None, fresh_unet(), RefMel(**MEL16), DDIMScheduler())
    gl.INIT_PHASES[:] = list(ph)
    kw = dict(raw_audio=raw, slice=1, steps=6, mask_start_secs=0.02, mask_end_secs=0.03)
    noise_io = noise[:1].clone()
    images, _ = pipe(batch_size=1, noise=noise_io, start_step=2, return_dict=False, **kw)
    out.update({"D:raw": raw, "D:noise": noise.numpy(), "D:images": u8(images), "D:noise_after": noise_io.numpy(),
                "D:cond_image": np.asarray(pipe.mel.audio_slice_to_image(1)), "D:slices": np.array(pipe.mel.get_number_of_slices())})
    gl.INIT_PHASES[:] = list(ph)
    images, _ = pipe(batch_size=1, noise=noise[1:].clone(), start_step=0, return_dict=False, **kw)
    out["D2:images"] = u8(images)
    for key, ss in (("D3:raises", 2), ("D4:raises", 0)):    # two rows: neither the start-step write nor the mask write broadcasts
        try:
            pipe(batch_size=2, noise=noise.clone(), start_step=ss, return_dict=False, **kw)
            out[key] = np.array("")
        except RuntimeError as e:
            out[key] = np.array(type(e).__name__)

    # ---- E: latent diffusion: AutoencoderKL encode(...).latent_dist.sample(generator) -> loop -> decode
    torch.manual_seed(5)
    vae = AutoencoderKL(**VAE_CFG).eval()
    for k, v in vae.state_dict().items():
        out["vae:" + k] = v.numpy()
    raw32 = pcm(0.2 * rs.standard_normal(MEL32["x_res"] * MEL32["hop_length"] + 17))
    noise = torch.randn(2, 1, 16, 16, generator=g)
    ph32 = phases(rs, MEL32, 2)
    pipe = RefPipeline(vae, fresh_unet(), RefMel(**MEL32), DDIMScheduler())
    gl.INIT_PHASES[:] = list(ph32)
    images, (_, audios) = pipe(batch_size=2, steps=3, noise=noise.clone(), return_dict=False)
    out.update({"E:noise": noise.numpy(), "E:phase": ph32.astype(np.float32), "E:images": u8(images), "E:audios": np.stack(audios)})
    gl.INIT_PHASES[:] = list(ph32)
    images, _ = pipe(batch_size=1, steps=3, noise=noise[:1].clone(), raw_audio=raw32, start_step=1, mask_end_secs=0.01,
                     generator=torch.Generator().manual_seed(9), return_dict=False)
    out.update({"E:raw": raw32, "E:images_from_audio": u8(images)})

    # ---- F: DDIM inversion of A's images, and slerp
    pipe = RefPipeline(None, fresh_unet(), RefMel(**MEL16), DDIMScheduler())
    enc = pipe.encode([__import__("PIL.Image").Image.fromarray(a) for a in out["A:images"]], steps=5)
    out["F:encoded"] = enc.numpy()
    x0, x1 = torch.randn(1, 16, 16, generator=g), torch.randn(1, 16, 16, generator=g)
    out.update({"F:x0": x0.numpy(), "F:x1": x1.numpy(), "F:slerp": RefPipeline.slerp(x0, x1, 0.3).numpy()})

    # ---- G: the Mel class on its own: padding of a short clip, slice count, both conversions
    mel = RefMel(**MEL32)
    short = pcm(0.25 * rs.standard_normal(1000))
    mel.load_audio(raw_audio=short)
    out.update({"G:short": short, "G:padded_len": np.array(len(mel.audio)), "G:padded_dtype": np.array(str(mel.audio.dtype)),
                "G:slices_short": np.array(mel.get_number_of_slices()), "G:image_short": np.asarray(mel.audio_slice_to_image(0))})
    mel.load_audio(raw_audio=raw32)
    img = mel.audio_slice_to_image(0)
    gl.INIT_PHASES[:] = [ph32[0]]
    out.update({"G:image": np.asarray(img), "G:audio": mel.image_to_audio(img), "G:slice_size": np.array(mel.slice_size),
                "G:sample_rate": np.array(mel.get_sample_rate())})

    # ---- J: conditional generation (:160-161: `self.unet(images, t, encoding)` when the model is a UNet2DConditionModel);
    # weights = torch.manual_seed(COND_SEED) + the oracle's constructor (digest in the fixture, not the tensors)
    cond = cond_unet()
    g = torch.Generator().manual_seed(8)
    noise = torch.randn(2, 1, 16, 16, generator=g)
    enc = torch.randn(2, 3, COND_CFG["cross_attention_dim"], generator=g)
    pipe = RefPipeline(None, cond, RefMel(**MEL16), DDIMScheduler())
    gl.INIT_PHASES[:] = list(ph)
    images, _ = pipe(batch_size=2, steps=4, noise=noise.clone(), encoding=enc, return_dict=False)
    out.update({"J:noise": noise.numpy(), "J:encoding": enc.numpy(), "J:images": u8(images),
                "J:sd_sha256": np.array(state_sha256(cond.state_dict()))})

    # ---- N: the long-form procedures = code cells of notebooks/audio_diffusion_pipeline.ipynb, exec'd as written (form fields
    # marked `#@param` may be overridden, as the notebook intends; `display` / `Audio` are no-ops)
    import librosa
    cells = notebook_cells()
    mel_n = RefMel(**MEL_NB)
    unet.sample_size = (MEL_NB["y_res"], MEL_NB["x_res"])
    pipe = RefPipeline(None, unet, mel_n, DDIMScheduler())
    ns = dict(audio_diffusion=pipe, mel=mel_n, sample_rate=mel_n.get_sample_rate(), display=lambda *a, **k: None,
              Audio=lambda *a, **k: None, np=np, torch=torch, librosa=librosa, device="cpu", generator=torch.Generator())
    phn = notebook_phases()

    class FixedSeedGenerator(torch.Generator):        # the remix cell draws its seed from the OS (`generator.seed()`)
        def seed(self):
            self.manual_seed(NB_REMIX_SEED)
            return NB_REMIX_SEED
    # cell "Generate continuations ('out-painting')": 12 segments, each pinned to the previous tail by mask_start_secs
    ns["audio"] = notebook_clip(1.0, 3)
    gl.INIT_PHASES[:] = [p.astype(np.float64) for p in phn[:12]]
    torch.manual_seed(NB_SEED)
    exec(cell_source(cells, "mask_start_secs=overlap_secs)", {}), ns)
    track = np.asarray(ns["track"], dtype=np.float32)
    out.update({"N:outpaint_len": np.array(len(track)), "N:outpaint_track_every4": track[::4].copy(),
                "N:outpaint_segment_l2": np.array([np.linalg.norm(seg.astype(np.float64)) for seg in np.array_split(track, 13)]),
                "N:outpaint_last_image": np.asarray(ns["image2"])})
    # cell "Remix (style transfer)": a file, overlapping slices from start_step, generated tail re-inserted into the next slice
    with tempfile.TemporaryDirectory() as tmp:
        ns["audio_file"] = os.path.join(tmp, "track.wav")
        notebook_wav(ns["audio_file"])
        gl.INIT_PHASES[:] = [p.astype(np.float64) for p in phn[12:12 + NB_REMIX_SLICES]]
        import types
        ns_remix = dict(ns, torch=types.SimpleNamespace(Generator=FixedSeedGenerator))
        exec(cell_source(cells, "not_first = 0", {"start_step": NB_START_STEP}), ns_remix)
        ns.update({k: ns_remix[k] for k in ("track", "track_audio", "stride", "seed", "audio2")})
    assert len(ns["track_audio"]) // ns["stride"] == NB_REMIX_SLICES and not gl.INIT_PHASES
    assert ns["seed"] == NB_REMIX_SEED
    out.update({"N:remix_track": np.asarray(ns["track"], dtype=np.float32)})
    # cells "Encode / reconstruct / interpolate": two images, DDIM inversion, reconstruction, slerp at alpha, sampling
    ima, imb = [Image.fromarray(a) for a in notebook_images()]
    ns["ds"] = {"train": {264: {"image": ima}, 15978: {"image": imb}}}
    for marker, ph_i in (("image = ds['train'][264]['image']", None), ("noise = audio_diffusion.encode([image])", None),
                         ("# Reconstruct original audio from noise", 16), ("image2 = ds['train'][15978]['image']", None),
                         ("noise2 = audio_diffusion.encode([image2])", None), ("audio_diffusion.slerp(noise, noise2, alpha)", 17)):
        gl.INIT_PHASES[:] = [] if ph_i is None else [phn[ph_i].astype(np.float64)]
        exec(cell_source(cells, marker, {}), ns)
        if marker.startswith("# Reconstruct"):
            out.update({"N:reconstructed_image": np.asarray(ns["image"]), "N:reconstructed_audio": np.asarray(ns["audio"], dtype=np.float32)})
    out.update({"N:noise": ns["noise"].numpy(), "N:noise2": ns["noise2"].numpy(), "N:alpha": np.array(ns["alpha"]),
                "N:slerp_audio": np.asarray(ns["audio"], dtype=np.float32), "N:slerp_image": np.asarray(ns["output"].images[0])})
    unet.sample_size = UNET_CFG["sample_size"]

    # ---- H: the AudioEncoder module (audio_encoder.py:62-84), eval mode as `encode` runs it (:87-88); 42 M weights, so the
    # fixture holds the seeds (oracle.audio_encoder.random_state_dict) and a digest instead of the state dict
    import hashlib
    from audiodiffusion.audio_encoder import AudioEncoder as RefEncoder
    from oracle import audio_encoder as oenc
    sd = oenc.random_state_dict(3)
    enc = RefEncoder()
    enc.load_state_dict(sd, strict=True)               # every key name and shape of the reference module
    enc.eval()
    x = torch.rand((2, 1, 96, 216), generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        y = enc(x)
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(sd[k].numpy().tobytes())
    out.update({"H:embedding": y.numpy(), "H:sd_seed": np.array(3), "H:x_seed": np.array(4), "H:sd_sha256": np.array(h.hexdigest()),
                "H:mel_res": np.array([enc.mel.x_res, enc.mel.y_res])})
    # AudioEncoder.encode (:86-107) as written: file -> every 216-frame slice -> image / 255 -> forward -> mean over the slices
    with tempfile.TemporaryDirectory() as tmp:
        wav = os.path.join(tmp, "clip.wav")
        encoder_wav(wav)
        out["H:encode_average"] = enc.encode([wav, wav]).numpy()
        try:                                            # pool="max" assigns the (values, indices) pair of torch.max and fails
            enc.encode([wav], pool="max")
            out["H:encode_max_raises"] = np.array("")
        except Exception as e:                          # noqa: BLE001
            out["H:encode_max_raises"] = np.array(type(e).__name__)

    # ---- I: the dataset builder script (scripts/audio_to_images.py main(), run as written) on three files: a chirp of 2.6
    # slices, a file whose first slice is digital silence (skipped, :48-51), and one the decoder rejects (reported and
    # skipped, :36-42).  Non-square resolution: width 32 = x_res frames, height 16 = mel bins.
    import argparse
    import importlib.util
    from datasets import load_from_disk
    spec = importlib.util.spec_from_file_location("reference_audio_to_images", os.path.join(REFERENCE, "scripts", "audio_to_images.py"))
    script = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(script)
    with tempfile.TemporaryDirectory() as tmp:
        wavs = dataset_wavs(os.path.join(tmp, "in"))
        args = argparse.Namespace(input_dir=os.path.join(tmp, "in"), output_dir=os.path.join(tmp, "out"), resolution=DATASET_RES,
                                  hop_length=128, push_to_hub=None, sample_rate=22050, n_fft=512)
        script.main(args)
        ds = load_from_disk(args.output_dir)["train"]
        rows = sorted(((os.path.relpath(r["audio_file"], args.input_dir), int(r["slice"]), np.asarray(r["image"])) for r in ds),
                      key=lambda r: r[:2])
        out.update({"I:files": np.array([r[0] for r in rows]), "I:slices": np.array([r[1] for r in rows], dtype=np.int16),
                    "I:images": np.stack([r[2] for r in rows]), "I:features": np.array(str(ds.features)),
                    "I:wav_sha256": np.array([wavs[k] for k in sorted(wavs)])})

    np.savez_compressed(out_path, **out)
    print({k: (v.shape, str(v.dtype)) for k, v in out.items() if not k.startswith("vae:")})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "reference_pipeline.npz"))
