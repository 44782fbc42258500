"""BASELINE.json's full sizes on the MI355X (gpu-only): the 113.67 M-parameter 256x256 UNet, DDIM-50, the config-4 VAE and
the default Mel codec. The oracle is affordable at batch 1 (seconds on the GPU box's host cores); beyond that the tests
use size-independent properties of the path: samples never interact (batch-shard invariance — the property the
multi-GPU sampling relies on), the loop is deterministic, the uint8 image is the documented rounding of the float image.
Tolerances: SURVEY.md §8(c) / north_star — 1e-3 in fp32 units, uint8 images within 1 LSB and >= 99.5 % identical."""
import numpy as np
import pytest
import torch

from native_backend import select

pytestmark = pytest.mark.gpu

CFG256 = dict(sample_size=(256, 256), in_channels=1, out_channels=1, layers_per_block=2,
              block_out_channels=(128, 128, 256, 256, 512, 512),
              down_block_types=("DownBlock2D",) * 4 + ("AttnDownBlock2D", "DownBlock2D"),
              up_block_types=("UpBlock2D", "AttnUpBlock2D") + ("UpBlock2D",) * 4)


def _record(what, **figures):
    """ADM_ACCURACY_LOG=<file>: every end-to-end figure these tests measure is appended there as one JSON line (with the `wino6` setting the
    library runs under) — `tools/r06_accuracy.sh` runs the file with the F(4x4) kernel on and off and formats profiles/r06_accuracy.md from it."""
    import json
    import os
    path = os.environ.get("ADM_ACCURACY_LOG")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps(dict(what=what, wino6=os.environ.get("ADM_WINO6", "default"), **figures)) + "\n")


@pytest.fixture(scope="module")
def dev():
    return select("hip")


@pytest.fixture(scope="module")
def unet(dev):
    from audiodiffusion import UNet2DModel
    return UNet2DModel(**CFG256).init_random(0)


def test_unet_256_matches_the_oracle_at_batch_1(dev, unet):
    from oracle.unet import UNet2DModel as OracleUNet
    ref = OracleUNet(**CFG256).eval()
    ref.load_state_dict(unet.state_dict())
    x = torch.randn(1, 1, 256, 256, generator=torch.Generator().manual_seed(42))
    for t in (980, 20):
        with torch.no_grad():
            r = ref(x, torch.tensor(t))["sample"]
        o = unet(x.to(dev), torch.tensor(t))["sample"].cpu()
        _record(f"whole UNet 256x256, B = 1, t = {t}: max|d| / max|ref|", value=float((o - r).abs().max()) / float(r.abs().max()), bar=1e-4)
        assert float((o - r).abs().max()) <= 1e-4 * float(r.abs().max()), (t, float((o - r).abs().max()))


def test_unet_256_samples_do_not_interact(dev, unet):
    """Forward of a batch == forwards of its rows (what row-sharding a global batch over GPUs relies on); per-sample timesteps."""
    x = torch.randn(5, 1, 256, 256, generator=torch.Generator().manual_seed(1)).to(dev)
    ts = torch.tensor([0, 37, 500, 980, 999])
    full = unet(x, ts)["sample"]
    for i in (0, 3, 4):
        one = unet(x[i:i + 1].contiguous(), ts[i:i + 1])["sample"]
        assert float((full[i:i + 1] - one).abs().max()) <= 1e-6 * float(one.abs().max())
    assert torch.equal(unet(x, ts)["sample"], full), "the forward is not deterministic"


def test_bench_batch_rows_are_bit_identical_to_single_sample_runs(dev, unet):
    """The bench batch (B = 32, config 3's per-GPU shard) against single-sample runs, BIT FOR BIT (SURVEY §8(c) anchor 7):
    rows 0 / 17 / 31 of one forward equal forwards of those samples alone, and row 5 of a complete DDIM-50 sampling at B = 32 (captured
    hipGraph) equals the same start noise sampled alone.  This is what lets the single B = 1 comparison against the oracle speak for the
    bench batch and for every multi-GPU shard: no kernel's summation order, split or tile walk depends on the batch it runs in."""
    from audiodiffusion import AudioDiffusionPipeline, DDIMScheduler, Mel
    B = 32
    x = torch.randn(B, 1, 256, 256, generator=torch.Generator().manual_seed(11)).to(dev)
    ts = torch.tensor([(37 * i) % 1000 for i in range(B)])
    full = unet(x, ts)["sample"]
    for r in (0, 17, 31):
        one = unet(x[r:r + 1].contiguous(), ts[r:r + 1])["sample"]
        assert torch.equal(one[0], full[r]), (r, float((one[0] - full[r]).abs().max()))
    pipe = AudioDiffusionPipeline(None, unet, Mel(), DDIMScheduler())
    pipe.set_progress_bar_config(disable=True)
    noise = torch.randn(B, 1, 256, 256, generator=torch.Generator().manual_seed(12)).to(dev)
    imgs, flt = pipe(batch_size=B, noise=noise.clone(), audio=False, return_float=True)          # 50 steps at the bench batch
    i1, f1 = pipe(batch_size=1, noise=noise[5:6].clone(), audio=False, return_float=True)
    assert torch.equal(f1[0], flt[5]), float((f1[0] - flt[5]).abs().max())
    assert np.array_equal(np.asarray(i1[0]), np.asarray(imgs[5]))


def test_ddim50_loop_is_deterministic_shardable_and_quantises_as_documented(dev, unet):
    from audiodiffusion import AudioDiffusionPipeline, DDIMScheduler, Mel
    pipe = AudioDiffusionPipeline(None, unet, Mel(), DDIMScheduler())
    pipe.set_progress_bar_config(disable=True)
    noise = torch.randn(3, 1, 256, 256, generator=torch.Generator().manual_seed(42)).to(dev)
    imgs, flt = pipe(batch_size=3, noise=noise.clone(), audio=False, return_float=True)        # 50 steps, hipGraph
    assert pipe.get_default_steps() == 50 and len(imgs) == 3 and tuple(flt.shape) == (3, 1, 256, 256)
    imgs2, flt2 = pipe(batch_size=3, noise=noise.clone(), audio=False, return_float=True)
    assert torch.equal(flt, flt2), "two identical samplings differ"
    # shard invariance: rows sampled alone / in a different batch give the same spectrogram
    i1, f1 = pipe(batch_size=1, noise=noise[1:2].clone(), audio=False, return_float=True)
    assert float((f1 - flt[1:2]).abs().max()) <= 1e-4
    a, b = np.asarray(i1[0]).astype(int), np.asarray(imgs[1]).astype(int)
    assert np.abs(a - b).max() <= 1 and (a == b).mean() >= 0.995
    # uint8 image = round-half-even((x/2+0.5).clamp(0,1)*255) of the float image (pipeline_audio_diffusion.py:192-197)
    want = ((flt / 2 + 0.5).clamp(0, 1) * 255).round().to(torch.uint8).cpu().numpy()[:, 0]
    got = np.stack([np.asarray(i) for i in imgs])
    assert got.shape == want.shape and np.abs(got.astype(int) - want.astype(int)).max() <= 1
    assert (got == want).mean() >= 0.9999
    assert np.isfinite(flt.cpu().numpy()).all() and float(flt.abs().max()) <= 1.0 + 1e-6      # clip_sample keeps x0 in [-1,1]


# ---------------------------------------------------------------------------------------------------------------------
# Loop-level parity AT BASELINE SIZES against the CPU oracle driven through the same procedure
# (`pipeline_audio_diffusion.py:159-199`): same weights, same start noise, same injected per-step scheduler noise.
# Bars: final float image <= 1e-3 (north_star), uint8 image <= 1 LSB and >= 99.5 % identical.
def _loop_parity(dev, cfg, sched_name, steps, start_step, B, seed, vae_cfg=None, eta=0.0, options=None):
    import os
    from audiodiffusion import AudioDiffusionPipeline, DDIMScheduler, DDPMScheduler, Mel, UNet2DModel
    from audiodiffusion.vae import AutoencoderKL
    from oracle import mel as omel
    from oracle import pipeline as opipe
    from oracle import schedulers as osched
    from oracle.unet import UNet2DModel as OracleUNet
    from oracle.vae import AutoencoderKL as OracleVAE
    torch.set_num_threads(min(32, os.cpu_count() or 8))            # the oracle convolutions do not scale past this
    unet = UNet2DModel(**cfg).init_random(seed)
    for k, v in (options or {}).items():                           # per-MODEL options (adm_unet_set_option)
        unet.set_option(k, v)
    ref_unet = OracleUNet(**cfg).eval()
    ref_unet.load_state_dict(unet.state_dict())
    vae = ref_vae = None
    if vae_cfg is not None:
        vae = AutoencoderKL(**vae_cfg).init_random(seed + 1)
        ref_vae = OracleVAE(**vae_cfg).eval()
        ref_vae.load_state_dict(vae.state_dict())
    mine = AudioDiffusionPipeline(vae, unet, Mel(), (DDIMScheduler if sched_name == "ddim" else DDPMScheduler)()).to(dev)
    mine.set_progress_bar_config(disable=True)
    ref = opipe.AudioDiffusionPipeline(ref_vae, ref_unet, omel.Mel(),
                                       (osched.DDIMScheduler if sched_name == "ddim" else osched.DDPMScheduler)())
    g = torch.Generator().manual_seed(1000 + seed)
    hw = cfg["sample_size"]
    noise = torch.randn((B, cfg["in_channels"]) + tuple(hw), generator=g)
    n_run = steps - start_step
    step_noise = torch.randn((n_run, B, cfg["in_channels"]) + tuple(hw), generator=g)
    kw = dict(batch_size=B, steps=steps, start_step=start_step, eta=eta, audio=False, return_float=True)
    ri, rf = ref(noise=noise.clone(), step_noise=step_noise, **kw)
    mi, mf = mine(noise=noise.clone().to(dev), step_noise=step_noise.to(dev), **kw)
    err = float((mf.cpu() - rf).abs().max())
    a = np.stack([np.asarray(i).astype(int) for i in mi])
    b = np.stack([np.asarray(i).astype(int) for i in ri])
    assert a.shape == b.shape
    _record(f"loop {sched_name.upper()} {hw[0]}x{hw[1]}{' latent + VAE decode' if vae_cfg else ''}, steps {start_step}..{steps}, B = {B}, eta {eta}"
            + (f", model options {options}" if options else ""),
            float_err=err, bar=1e-3, max_lsb=int(np.abs(a - b).max()), identical_pixels=float((a == b).mean()))
    assert err <= 1e-3, err
    assert np.abs(a - b).max() <= 1 and (a == b).mean() >= 0.995, (np.abs(a - b).max(), (a == b).mean())
    return err


def test_config3_ddim50_single_steps_along_the_product_trajectory_match_the_oracle(dev):
    """configs[2] on the FULL 50-step schedule: with random weights a 50-step sampler amplifies rounding differences
    exponentially (tests/test_pipeline.py documents the rate on the oracle itself), so the whole trajectory of two fp32
    implementations cannot agree to 1e-3 — but every step can.  The product runs its captured loop to step k (k = 0, 9, 24,
    39, 49: early, middle, late, last), then BOTH sides take that one step from the product's state x_k: native loop vs
    `unet(x, t)` + `scheduler.step` of the oracle at 256x256.  Bar: 1e-4 per step (north_star's 1e-3 over the loop)."""
    import os
    from audiodiffusion import AudioDiffusionPipeline, DDIMScheduler, Mel, UNet2DModel
    from oracle import schedulers as osched
    from oracle.unet import UNet2DModel as OracleUNet
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    unet = UNet2DModel(**CFG256).init_random(4)
    ref_unet = OracleUNet(**CFG256).eval()
    ref_unet.load_state_dict(unet.state_dict())
    ref_sched = osched.DDIMScheduler()
    ref_sched.set_timesteps(50)
    mine = AudioDiffusionPipeline(None, unet, Mel(), DDIMScheduler()).to(dev)
    mine.set_progress_bar_config(disable=True)
    mine.scheduler.set_timesteps(50)
    x = torch.randn(1, 1, 256, 256, generator=torch.Generator().manual_seed(1004)).to(dev)
    k_prev, worst = 0, 0.0
    with torch.no_grad():
        for k in (0, 9, 24, 39, 49):
            if k > k_prev:
                x, _ = mine._denoise(x, k_prev, 0.0, None, None, 0, 0, stop_step=k)      # the product's own trajectory to x_k
            got, _ = mine._denoise(x, k, 0.0, None, None, 0, 0, stop_step=k + 1)
            t = ref_sched.timesteps[k]
            xc = x.cpu()
            want = ref_sched.step(model_output=ref_unet(xc, t)["sample"], timestep=t, sample=xc, eta=0.0)["prev_sample"]
            worst = max(worst, float((got.cpu() - want).abs().max()))
            x, k_prev = got, k + 1
    assert worst <= 1e-4, worst


def test_config3_complete_ddim50_sampling_matches_the_oracle_on_a_briefly_trained_model(dev):
    """BASELINE.json's headline workload END TO END (`pipeline_audio_diffusion.py:69,159-185`): a complete DDIM-50 sampling at
    256x256 from pure noise through the captured hipGraph against 50 oracle steps on the host cores, at the path's bars
    (<= 1e-3 float, <= 1 LSB, >= 99.5 % identical).  The denoiser is the 113.67 M-parameter model after a brief run of the
    product's OWN trainer (bf16 operands, fp32 masters; tests/brief_training.py): a trained denoiser is contractive, whereas
    random weights make the sampler chaotic for ANY pair of fp32 implementations — asserted below on the product against
    itself, and recorded at every checkpoint in profiles/r03_ddim50_parity.md."""
    import os
    from audiodiffusion import AudioDiffusionPipeline, DDIMScheduler, Mel, UNet2DModel
    from brief_training import train_product
    from oracle import mel as omel
    from oracle import pipeline as opipe
    from oracle import schedulers as osched
    from oracle.unet import UNet2DModel as OracleUNet
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    g = torch.Generator().manual_seed(1234)
    noise = torch.randn(1, 1, 256, 256, generator=g)
    both = torch.cat([noise, noise + 1e-6 * torch.randn(1, 1, 256, 256, generator=g)])

    def sample(unet, x):
        pipe = AudioDiffusionPipeline(None, unet, Mel(), DDIMScheduler()).to(dev)
        pipe.set_progress_bar_config(disable=True)
        return pipe(batch_size=x.shape[0], noise=x.clone().to(dev), audio=False, return_float=True)       # steps=None -> 50

    # random weights: the sampler amplifies a 1e-6 perturbation of the start noise beyond the path's bar all by itself
    _, f = sample(UNet2DModel(**CFG256).init_random(0), both)
    chaos = float((f[0] - f[1]).abs().max())
    assert chaos > 1e-3, chaos
    trainee = UNet2DModel(**CFG256).init_random(0)
    losses = train_product(trainee, (256, 256), 120, dev, batch=16, lr=1e-4)
    assert np.isfinite(losses).all() and np.mean(losses[-10:]) < 0.5 * np.mean(losses[:3]), (losses[:3], losses[-10:])
    sd = trainee.state_dict()
    del trainee
    unet = UNet2DModel(**CFG256).load_state_dict(sd)
    mi, mf = sample(unet, both)
    calm = float((mf[0] - mf[1]).abs().max())
    assert calm <= 1e-4, calm                                # contractive: the perturbation does not grow any more
    ref_unet = OracleUNet(**CFG256).eval()
    ref_unet.load_state_dict(sd)
    ref = opipe.AudioDiffusionPipeline(None, ref_unet, omel.Mel(), osched.DDIMScheduler())
    ri, rf = ref(batch_size=1, noise=noise.clone(), audio=False, return_float=True)                        # 50 oracle steps
    err = float((mf[0:1].cpu() - rf).abs().max())
    a, b = np.asarray(mi[0]).astype(int), np.asarray(ri[0]).astype(int)
    _record("complete DDIM-50 256x256, checkpoint 1 (120 optimizer steps), noise seed 1234", float_err=err, bar=1e-3,
            max_lsb=int(np.abs(a - b).max()), identical_pixels=float((a == b).mean()), perturbation_growth=calm)
    assert err <= 1e-3, err
    assert np.abs(a - b).max() <= 1 and (a == b).mean() >= 0.995, (np.abs(a - b).max(), (a == b).mean())
    assert float(rf.std()) > 0.02, "degenerate sample: the comparison would be vacuous"
    # A SECOND checkpoint (40 more optimizer steps of the same run) and a SECOND start noise (the headline's only
    # end-to-end oracle comparison was one seed on one checkpoint): the same three bars.
    trainee = UNet2DModel(**CFG256).load_state_dict(sd)
    losses2 = train_product(trainee, (256, 256), 40, dev, batch=16, lr=1e-4, seed=3)
    assert np.isfinite(losses2).all()
    sd2 = trainee.state_dict()
    del trainee
    assert max(float((sd2[k].float() - sd[k].float()).abs().max()) for k in sd if sd[k].dtype.is_floating_point) > 0     # it did move
    noise2 = torch.randn(1, 1, 256, 256, generator=torch.Generator().manual_seed(777))
    mi2, mf2 = sample(UNet2DModel(**CFG256).load_state_dict(sd2), noise2)
    ref_unet.load_state_dict(sd2)
    ri2, rf2 = ref(batch_size=1, noise=noise2.clone(), audio=False, return_float=True)
    err2 = float((mf2.cpu() - rf2).abs().max())
    a2, b2 = np.asarray(mi2[0]).astype(int), np.asarray(ri2[0]).astype(int)
    _record("complete DDIM-50 256x256, checkpoint 2 (+40 optimizer steps), noise seed 777", float_err=err2, bar=1e-3,
            max_lsb=int(np.abs(a2 - b2).max()), identical_pixels=float((a2 == b2).mean()))
    assert err2 <= 1e-3, err2
    assert np.abs(a2 - b2).max() <= 1 and (a2 == b2).mean() >= 0.995, (np.abs(a2 - b2).max(), (a2 == b2).mean())
    assert float(rf2.std()) > 0.02 and float((rf2 - rf).abs().max()) > 0.05, "the second sample must be another picture"


def test_config2_ddpm_1000_schedule_crosses_a_noise_staging_chunk_boundary_at_size(dev):
    """configs[1], the part the last-4-steps test cannot reach: the DDPM noise-staging path of the native loop
    (`_STEP_CHUNK`: per-step variance noise staged chunk by chunk, a stream sync and a rewritten staging buffer at every
    boundary, the captured graph replayed on the next chunk).  Steps 895-905 of the 1000-step schedule at 256x256 with
    injected noise and a 5-step chunk, so the boundary falls at step 900 — the same code a full sampling crosses nine times."""
    import os
    from audiodiffusion import AudioDiffusionPipeline, DDPMScheduler, Mel, UNet2DModel
    from oracle import schedulers as osched
    from oracle.unet import UNet2DModel as OracleUNet
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    unet = UNet2DModel(**CFG256).init_random(6)
    ref_unet = OracleUNet(**CFG256).eval()
    ref_unet.load_state_dict(unet.state_dict())
    mine = AudioDiffusionPipeline(None, unet, Mel(), DDPMScheduler()).to(dev)
    mine.set_progress_bar_config(disable=True)
    mine.scheduler.set_timesteps(1000)
    mine._STEP_CHUNK = 5                                     # instance override: boundary after 5 of the 10 steps
    s = osched.DDPMScheduler()
    s.set_timesteps(1000)
    g = torch.Generator().manual_seed(1006)
    x0 = 0.3 * torch.randn(1, 1, 256, 256, generator=g)      # a late-trajectory magnitude
    step_noise = torch.randn(10, 1, 1, 256, 256, generator=g)
    got, _ = mine._denoise(x0.to(dev), 895, 0.0, None, None, 0, 0, step_noise=step_noise.to(dev), stop_step=905)
    whole = AudioDiffusionPipeline(None, unet, Mel(), DDPMScheduler()).to(dev)
    whole.scheduler.set_timesteps(1000)
    one, _ = whole._denoise(x0.to(dev), 895, 0.0, None, None, 0, 0, step_noise=step_noise.to(dev), stop_step=905)
    assert torch.equal(got, one), "chunked and unchunked staging differ"
    x = x0.clone()
    with torch.no_grad():
        for i, t in enumerate(s.timesteps[895:905]):
            x = s.step(ref_unet(x, t)["sample"], t, x, variance_noise=step_noise[i])["prev_sample"]
    err = float((got.cpu() - x).abs().max())
    assert err <= 1e-3, err


def test_config1_64x64_ddpm_10_steps_loop_matches_the_oracle(dev):
    """BASELINE.json configs[0]: audio-diffusion-64 (the train_unet.py architecture at 64x64), DDPM, 1 sample, 10 steps —
    the noise term of every DDPM step is injected so both sides consume identical draws."""
    cfg = dict(CFG256, sample_size=(64, 64))
    _loop_parity(dev, cfg, "ddpm", 10, 0, 1, seed=0)


def test_config1_64x64_ddpm_10_steps_under_the_single_sample_rules(dev):
    """configs[0] as bench.py times it: the model carries `set_option("single_sample", 1)` (split-K partitions for layers whose tiles cannot
    fill the chip with one sample — what the AudioDiffusion front end selects). Same bar as under the default rules."""
    cfg = dict(CFG256, sample_size=(64, 64))
    _loop_parity(dev, cfg, "ddpm", 10, 0, 1, seed=0, options={"single_sample": 1})


def test_unet_256_under_the_single_sample_front_end_rules(dev, unet):
    """The 256x256 model under the two per-model rules `AudioDiffusion` sets ("wino6" = 256, "single_sample" = 1): one forward against the
    oracle at the whole-network bar, and — the partitions are functions of the layer, not of the batch — rows of a batch bit-identical to
    the same samples alone. (The default-rule model of this module is untouched: the options live on the model handle.)"""
    from audiodiffusion import UNet2DModel
    from oracle.unet import UNet2DModel as OracleUNet
    mine = UNet2DModel(**CFG256)
    mine.load_state_dict(unet.state_dict())
    mine.set_option("wino6", 256).set_option("single_sample", 1)
    ref = OracleUNet(**CFG256).eval()
    ref.load_state_dict(unet.state_dict())
    x = torch.randn(3, 1, 256, 256, generator=torch.Generator().manual_seed(43))
    ts = torch.tensor([980, 500, 20])
    with torch.no_grad():
        r = ref(x[:1], ts[0])["sample"]
    full = mine(x.to(dev), ts)["sample"]
    err = float((full[:1].cpu() - r).abs().max()) / float(r.abs().max())
    _record("whole UNet 256x256, B = 1, t = 980, model options wino6 = 256 + single_sample = 1: max|d| / max|ref|", value=err, bar=1e-4)
    assert err <= 1e-4, err
    for i in range(3):
        one = mine(x[i:i + 1].to(dev), ts[i:i + 1])["sample"]
        assert torch.equal(one[0], full[i]), (i, float((one[0] - full[i]).abs().max()))
    assert not torch.equal(unet(x[:1].to(dev), ts[:1])["sample"], full[:1])      # the default-rule model rounds differently (and was not touched)


def test_config2_256_ddpm_1000_schedule_last_steps_match_the_oracle(dev):
    """configs[1]: 256x256 pixel-space DDPM on the 1000-step schedule; the oracle affords the last 4 of them at batch 2
    (start_step = 996: timesteps 3, 2, 1, 0 — three with the injected variance noise, the last one without)."""
    _loop_parity(dev, CFG256, "ddpm", 1000, 996, 2, seed=1)


def test_config3_256_ddim_10_steps_loop_matches_the_oracle(dev):
    """configs[2]: 256x256 DDIM (eta = 0), a complete 10-step sampling from pure noise through the captured hipGraph."""
    _loop_parity(dev, CFG256, "ddim", 10, 0, 1, seed=2)


def test_config3_256_ddim_eta_1_uses_the_injected_noise(dev):
    """DDIM with eta = 1 draws variance noise every step (`pipeline_audio_diffusion.py:165-172`): 4 steps, batch 1."""
    _loop_parity(dev, CFG256, "ddim", 4, 0, 1, seed=3, eta=1.0)


def test_config4_latent_ddpm_10_steps_and_vae_decode_match_the_oracle(dev):
    """configs[3]: latent audio diffusion — DDPM-10 on 1x32x32 latents with the train_unet.py architecture, then
    AutoencoderKL.decode(latents / 0.18215) to 256x256 and the uint8 conversion (`pipeline_audio_diffusion.py:187-199`)."""
    vae_cfg = dict(sample_size=(256, 256), in_channels=1, out_channels=1, latent_channels=1, layers_per_block=2,
                   block_out_channels=(128, 256, 512, 512), down_block_types=("DownEncoderBlock2D",) * 4,
                   up_block_types=("UpDecoderBlock2D",) * 4)
    cfg = dict(CFG256, sample_size=(32, 32))
    _loop_parity(dev, cfg, "ddpm", 10, 0, 1, seed=4, vae_cfg=vae_cfg)


def test_training_step_256_gradients_match_autograd_at_batch_1(dev):
    import torch.nn.functional as F
    from audiodiffusion import UNet2DModel
    from oracle.unet import UNet2DModel as OracleUNet
    m = UNet2DModel(**CFG256).init_random(0)
    ref = OracleUNet(**CFG256)
    ref.load_state_dict(m.state_dict())
    flat, grads = m.enable_training()
    g = torch.Generator().manual_seed(9)
    x, tgt = torch.randn(1, 1, 256, 256, generator=g), torch.randn(1, 1, 256, 256, generator=g)
    ts = torch.tensor([321])
    lr = F.mse_loss(ref(x, ts)["sample"], tgt)
    lr.backward()
    lm = m.train_step(x.to(dev), ts, tgt.to(dev))
    assert abs(float(lm) - float(lr)) <= 1e-5 * float(lr)
    gmax = max(float(p.grad.abs().max()) for p in ref.parameters())
    worst = 0.0
    for name, p in ref.named_parameters():
        off = m.flat.offsets[name][0]
        got = grads[off:off + p.numel()].view(p.shape).cpu()
        worst = max(worst, float((got - p.grad).abs().max()) / max(float(p.grad.abs().max()), 1e-3 * gmax))
    assert worst <= 1e-3, worst


_BF16_ORACLE = {}


def _oracle_bf16_step(m):
    """fp32 autograd and torch.autocast(bfloat16) gradients of the oracle for one B = 1 step (computed once per session)."""
    if not _BF16_ORACLE:
        import torch.nn.functional as F
        from oracle.unet import UNet2DModel as OracleUNet
        ref = OracleUNet(**CFG256)
        ref.load_state_dict(m.state_dict())
        g = torch.Generator().manual_seed(9)
        x, tgt = torch.randn(1, 1, 256, 256, generator=g), torch.randn(1, 1, 256, 256, generator=g)
        ts = torch.tensor([321])
        lr = F.mse_loss(ref(x, ts)["sample"], tgt)
        lr.backward()
        g32 = {n: p.grad.clone() for n, p in ref.named_parameters()}
        ref.zero_grad()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            lac = F.mse_loss(ref(x, ts)["sample"].float(), tgt)
        lac.backward()
        gac = {n: p.grad.float().clone() for n, p in ref.named_parameters()}
        _BF16_ORACLE["v"] = (x, ts, tgt, lr.detach(), g32, lac.detach(), gac)
    return _BF16_ORACLE["v"]


@pytest.mark.parametrize("level,batch", [(3, 1), (3, 4), (2, 1)], ids=["3", "3-batch4", "2"])
def test_training_step_256_bf16_gradients_within_the_autocast_bars(dev, level, batch, monkeypatch):
    """BASELINE config 5 as written — 256x256, `--mixed_precision bf16` (scripts/train_unet.py:250-267, 391-401) — at size: ONE
    training step of the 113.67 M-parameter model at B = 1 against fp32 autograd of the oracle and against the reference's own
    mixed-precision mode (torch.autocast(bfloat16) on the oracle, what accelerate applies).  The toy model's three bars
    (tests/test_unet_training.py: global relative L2 error of the whole gradient < 1.5e-2, worst per-tensor error < 2e-2,
    no less accurate than autocast), for level 3 (blocked operand images, round 4) and level 2 (round 2's kernels).
    batch 4 = the same sample four times (the mean loss and its gradient are those of B = 1): the 16x16 and 8x8 levels then run the
    narrow-row tilings of the blocked kernels (2 / 4 images per tile, K split), which a single sample cannot fill."""
    from audiodiffusion import UNet2DModel, _native
    monkeypatch.setenv("ADM_BF16_LEVEL", str(level))
    m = UNet2DModel(**CFG256).init_random(0)
    x, ts, tgt, lr, g32, lac, gac = _oracle_bf16_step(m)
    gmax = max(float(v.abs().max()) for v in g32.values())

    def errs(get):
        num = den = worst = 0.0
        for n in g32:
            d = get(n) - g32[n]
            num += float(d.double().pow(2).sum())
            den += float(g32[n].double().pow(2).sum())
            worst = max(worst, float(d.abs().max()) / max(float(g32[n].abs().max()), 0.1 * gmax))
        return (num / den) ** 0.5, worst

    try:
        flat, grads = m.enable_training(mixed_precision="bf16")
        rep = lambda t: t.repeat((batch,) + (1,) * (t.dim() - 1))  # noqa: E731
        lm = float(m.train_step(rep(x).to(dev), rep(ts), rep(tgt).to(dev)))
        mine = errs(lambda n: grads[m.flat.offsets[n][0]:m.flat.offsets[n][0] + g32[n].numel()].view(g32[n].shape).cpu())
    finally:
        _native.check(_native.lib().adm_set_option(b"conv_bf16", 0))
    auto = errs(lambda n: gac[n])
    print(f"bf16 level {level} (B = {batch}) at 256x256: loss {lm:.6f} (fp32 {float(lr):.6f}, autocast {float(lac):.6f}); "
          f"gradient error global / worst tensor: product {mine[0]:.3e} / {mine[1]:.3e}, autocast {auto[0]:.3e} / {auto[1]:.3e}")
    assert abs(lm - float(lr)) <= 5e-3 * float(lr)
    assert mine[0] < 1.5e-2 and mine[1] < 2e-2, mine
    assert mine[0] <= auto[0], (mine, auto)
    assert mine[0] > 1e-4


def test_vae_256_matches_the_oracle_and_rows_are_independent(dev):
    from audiodiffusion.vae import AutoencoderKL
    from oracle.vae import AutoencoderKL as OracleVAE
    cfg = dict(sample_size=(256, 256), in_channels=1, out_channels=1, latent_channels=1, layers_per_block=2,
               block_out_channels=(128, 256, 512, 512), down_block_types=("DownEncoderBlock2D",) * 4,
               up_block_types=("UpDecoderBlock2D",) * 4)
    v = AutoencoderKL(**cfg).init_random(0)
    ref = OracleVAE(**cfg).eval()
    ref.load_state_dict(v.state_dict())
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 1, 256, 256, generator=g)
    with torch.no_grad():
        d = ref.encode(x[:1]).latent_dist
        nz = torch.randn(d.mean.shape, generator=g)
        rz = d.sample(noise=nz)
        rd = ref.decode(rz)["sample"]
    mz = v.encode(x[:1].to(dev)).latent_dist.sample(noise=nz.to(dev))
    md = v.decode(rz.to(dev))["sample"]
    assert float((mz.cpu() - rz).abs().max()) <= 1e-3 and float((md.cpu() - rd).abs().max()) <= 1e-3
    both = v.encode(x.to(dev)).latent_dist.mode()
    assert float((both[:1] - v.encode(x[:1].to(dev)).latent_dist.mode()).abs().max()) <= 1e-5


def test_conditional_unet_at_the_512_resolution_latent_size(dev):
    """SURVEY 8(f2) at size: `UNet2DConditionModel` as scripts/train_unet.py:139-159 builds it (135.6 M parameters), 64x64
    latents = the 512-resolution model — 4096 tokens per Transformer2DModel block at the first level, where self-attention runs as the flash
    kernel on the f32 MFMAs (round 5) — against the oracle at B = 1, and rows of a batch against single-sample runs, bit for bit."""
    from audiodiffusion import UNet2DConditionModel
    from oracle.unet_condition import UNet2DConditionModel as Oracle
    cfg = dict(sample_size=(64, 64), in_channels=1, out_channels=1, layers_per_block=2, block_out_channels=(128, 256, 512, 512),
               down_block_types=("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",),
               up_block_types=("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3, cross_attention_dim=100, attention_head_dim=8)
    m = UNet2DConditionModel(**cfg).init_random(0)
    ref = Oracle(**cfg).eval()
    ref.load_state_dict(m.state_dict())
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 1, 64, 64, generator=g)
    enc = torch.randn(3, 1, 100, generator=g)
    t = torch.tensor(321)
    with torch.no_grad():
        want = ref(x[:1], t, enc[:1])["sample"]
    got = m(x.to(dev), t, enc.to(dev))["sample"]
    assert float((got[:1].cpu() - want).abs().max()) <= 1e-4 * float(want.abs().max()), float((got[:1].cpu() - want).abs().max())
    for r in (0, 2):
        alone = m(x[r:r + 1].contiguous().to(dev), t, enc[r:r + 1].contiguous().to(dev))["sample"]
        assert torch.equal(alone[0], got[r])


def test_mel_default_config_forward_and_inverse_vs_oracle(dev):
    from audiodiffusion import Mel
    from oracle import mel as omel
    m, om = Mel(), omel.Mel()
    rng = np.random.default_rng(0)
    t = np.arange(m.slice_size * 3) / 22050.0
    y = (0.3 * rng.standard_normal(t.size) + 0.2 * np.sin(2 * np.pi * 440 * t) + 0.1 * np.sin(2 * np.pi * 3000 * t)).astype(np.float32)
    m.load_audio(raw_audio=y)
    om.load_audio(raw_audio=y)
    imgs = m.audio_slices_to_images([m.get_audio_slice(i) for i in range(3)])
    for i in range(3):       # same bar as tests/test_mel.py: <= 1 LSB and >= 99.9 % identical (FFT rounding at dB bin edges)
        a, b = imgs[i].astype(int), np.asarray(om.audio_slice_to_image(i)).astype(int)
        assert a.shape == b.shape == (256, 256)
        assert np.abs(a - b).max() <= 1 and (a == b).mean() >= 0.999, (np.abs(a - b).max(), (a == b).mean())
    phase = np.random.default_rng(1).random((1025, 256))
    img0 = om.audio_slice_to_image(0)                         # the SAME image and start phase for both inverses
    mine = m.images_to_audios([img0], init_phase=phase[None])[0]
    ref = om.image_to_audio(img0, init_phase=phase)
    assert mine.shape == ref.shape
    assert float(np.abs(mine - ref).max()) <= 1e-3 * max(1e-6, float(np.abs(ref).max()))


@pytest.mark.parametrize("shape", [(256, 128, 128, 128, 0, 0), (128, 0, 128, 128, 1, 0), (256, 256, 64, 256, 0, 1)],
                         ids=["concat-128px", "upsample-to-256px", "concat-64px-residual"])
def test_f4x4_layers_at_size_against_float64(dev, shape):
    """The reduced-accuracy path that ships: full-size layers of the 256x256 model through conv_wino6_kernel (Winograd
    F(4x4,3x3)) against a float64 reference on the host — GroupNorm(32) -> SiLU -> [nearest x2] -> 3x3 convolution + bias [+ residual].
    Measured 1.2-1.7e-5 of max|ref| on these three (profiles/r05_accuracy.md: the three largest of the table); the bar here is 4e-5 — a
    regression guard well inside the suite's per-layer 1e-4 — and the F(2x2) kernels on the same tensors stay below 3e-6."""
    from audiodiffusion import _native, ops
    lib = _native.lib()
    C1, C2, H, Co, up, has_res = shape
    C = C1 + C2
    g = torch.Generator().manual_seed(C1 + 7 * C2 + H + 3 * up)
    x = torch.randn(1, C, H, H, generator=g) * 1.5 + 0.3
    w = torch.randn(Co, C, 3, 3, generator=g) * (C * 9) ** -0.5
    b = torch.randn(Co, generator=g) * 0.1
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    Ho = 2 * H if up else H
    res = torch.randn(1, Co, Ho, Ho, generator=g) if has_res else None
    h = torch.nn.functional.silu(torch.nn.functional.group_norm(x.double(), 32, gamma.double(), beta.double(), 1e-5))
    if up:
        h = torch.nn.functional.interpolate(h, scale_factor=2.0, mode="nearest")
    ref = torch.nn.functional.conv2d(h, w.double(), b.double(), padding=1)
    if has_res:
        ref = ref + res.double()
    x1 = x[:, :C1].contiguous().to(dev)
    x2 = x[:, C1:].contiguous().to(dev) if C2 else None
    wd = w.to(dev)
    wp, wu = ops.pack_conv_weight(wd), ops.pack_winograd_weight(wd)
    gn = ops.groupnorm_stats(x1, gamma.to(dev), beta.to(dev), 32, 1e-5, x2=x2)
    try:
        for v6, bar in ((1, 4e-5), (0, 3e-6)):
            _native.check(lib.adm_set_option(b"wino6", v6))
            out = ops.conv2d(x1, wp, b.to(dev), 3, x2=x2, up=bool(up), gn=gn, act=True, wino=wu, residual=None if res is None else res.to(dev))
            assert (lib.adm_last_conv_variant() == 4316) == bool(v6)          # the default layer rule takes all three shapes
            err = float((out.double().cpu() - ref).abs().max() / ref.abs().max())
            assert err <= bar, (v6, err)
    finally:
        _native.check(lib.adm_set_option(b"wino6", -1))
