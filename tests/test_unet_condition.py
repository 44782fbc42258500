"""UNet2DConditionModel (scripts/train_unet.py:139-159; called at pipeline_audio_diffusion.py:160-161 with `encoding`):
native executor vs the oracle restatement on identical weights / inputs."""
import pytest
import torch

from native_backend import BACKENDS, select
from oracle.unet_condition import UNet2DConditionModel as OracleCond

TINY = dict(sample_size=16, in_channels=1, out_channels=1, layers_per_block=1, block_out_channels=(32, 64),
            down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"),
            cross_attention_dim=12, attention_head_dim=4)
TINY3 = dict(sample_size=(8, 16), in_channels=2, out_channels=2, layers_per_block=2, block_out_channels=(32, 32, 64),
             down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
             up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"), cross_attention_dim=20,
             attention_head_dim=8)


def test_reference_config_parameter_count():
    """Analytic anchor of the restatement: scripts/train_unet.py:139-159 with a 100-d encoding, 1 in/out channel."""
    m = OracleCond()
    n = sum(p.numel() for p in m.parameters())

    def resnet(ci, co):
        return 2 * ci + ci * co * 9 + co + 512 * co + co + 2 * co + co * co * 9 + co + (ci * co + co if ci != co else 0)

    def tr(c, d=100):
        return (2 * c + c * c + c) + 3 * 2 * c + (3 * c * c + c * c + c) + (c * c + 2 * c * d + c * c + c) \
            + (c * 8 * c + 8 * c + 4 * c * c + c) + (c * c + c)

    boc = (128, 256, 512, 512)
    want = 1 * 128 * 9 + 128 + (128 * 512 + 512) + (512 * 512 + 512)
    o = boc[0]
    for i, c in enumerate(boc):
        ci, o = o, c
        want += resnet(ci, o) + resnet(o, o) + (2 * tr(o) if i < 3 else 0) + (o * o * 9 + o if i < 3 else 0)
    want += 2 * resnet(512, 512) + tr(512)
    rev = boc[::-1]
    o = rev[0]
    for i, c in enumerate(rev):
        prev, o = o, c
        ci = rev[min(i + 1, 3)]
        for j in range(3):
            want += resnet((prev if j == 0 else o) + (ci if j == 2 else o), o)
        want += (3 * tr(o) if i > 0 else 0) + (o * o * 9 + o if i < 3 else 0)
    want += 2 * 128 + 128 * 9 + 1
    assert n == want == 135559809


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("cfg,B,S", [(TINY, 2, 1), (TINY3, 3, 4)], ids=["tiny-seq1", "tiny3-seq4"])
def test_conditional_unet_forward_matches_oracle(backend, cfg, B, S):
    dev = select(backend)
    from audiodiffusion.unet import UNet2DConditionModel
    torch.manual_seed(0)
    ref = OracleCond(**cfg).eval()
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "norm" in n:
                p.add_(0.2 * torch.randn_like(p))
    mine = UNet2DConditionModel(**cfg).load_state_dict(ref.state_dict())
    assert mine.num_parameters() == sum(p.numel() for p in ref.parameters())
    ss = cfg["sample_size"]
    hw = (ss, ss) if isinstance(ss, int) else ss
    g = torch.Generator().manual_seed(1)
    x = torch.randn((B, cfg["in_channels"]) + tuple(hw), generator=g)
    enc = torch.randn((B, S, cfg["cross_attention_dim"]), generator=g)
    ts = torch.tensor([5, 500, 999][:B])
    with torch.no_grad():
        want = ref(x, ts, enc)["sample"]
    got = mine(x.to(dev), ts, enc.to(dev))["sample"].cpu()
    assert got.shape == want.shape
    assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())
    # the encoding matters, and a new one is picked up by the next call
    enc2 = enc + 1.0
    with torch.no_grad():
        want2 = ref(x, ts, enc2)["sample"]
    got2 = mine(x.to(dev), ts, enc2.to(dev))["sample"].cpu()
    assert float((want2 - want).abs().max()) > 1e-3 * float(want.abs().max())
    assert float((got2 - want2).abs().max()) <= 1e-4 * float(want2.abs().max())
    with pytest.raises(ValueError):
        mine(x.to(dev), ts, None)


MEL = dict(x_res=16, y_res=16, hop_length=64, n_fft=256, n_iter=2, sample_rate=4000)


@pytest.mark.parametrize("backend", BACKENDS)
def test_conditional_pipeline_sampling_matches_oracle_and_roundtrips(backend, tmp_path):
    """`pipeline(..., encoding=...)` (pipeline_audio_diffusion.py:86,160-161): native loop with the encoding held constant
    over the steps vs the oracle loop; then save_pretrained / from_pretrained keep the class (`model_index.json`)."""
    import numpy as np
    from oracle import mel as omel
    from oracle import pipeline as opipe
    from oracle import schedulers as osched
    dev = select(backend)
    from audiodiffusion import AudioDiffusionPipeline, DDIMScheduler, Mel, UNet2DConditionModel
    torch.manual_seed(0)
    ref_unet = OracleCond(**TINY).eval()
    unet = UNet2DConditionModel(**TINY).load_state_dict(ref_unet.state_dict())
    ref = opipe.AudioDiffusionPipeline(None, ref_unet, omel.Mel(**MEL), osched.DDIMScheduler())
    mine = AudioDiffusionPipeline(None, unet, Mel(**MEL), DDIMScheduler())
    mine.set_progress_bar_config(disable=True)
    g = torch.Generator().manual_seed(42)
    noise = torch.randn(2, 1, 16, 16, generator=g)
    enc = torch.randn(2, 1, 12, generator=g)
    ri, rf = ref(batch_size=2, steps=4, noise=noise.clone(), encoding=enc, audio=False, return_float=True)
    mi, mf = mine(batch_size=2, steps=4, noise=noise.clone().to(dev), encoding=enc.to(dev), audio=False, return_float=True)
    assert float((mf.cpu() - rf).abs().max()) <= 1e-3
    a = np.stack([np.asarray(i).astype(int) for i in mi])
    b = np.stack([np.asarray(i).astype(int) for i in ri])
    assert np.abs(a - b).max() <= 1 and (a == b).mean() >= 0.995
    with pytest.raises(ValueError):
        mine(batch_size=2, steps=2, noise=noise.clone().to(dev), audio=False)         # conditional model, no encoding
    mine.save_pretrained(str(tmp_path / "cond"))
    again = AudioDiffusionPipeline.from_pretrained(str(tmp_path / "cond"))
    assert type(again.unet).__name__ == "UNet2DConditionModel" and again.unet.config.cross_attention_dim == 12
    again.set_progress_bar_config(disable=True)
    _, af = again(batch_size=2, steps=4, noise=noise.clone().to(dev), encoding=enc.to(dev), audio=False, return_float=True)
    assert torch.equal(af.cpu(), mf.cpu())


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("cfg,B,S", [(TINY, 2, 1), (TINY3, 3, 4)], ids=["tiny-seq1", "tiny3-seq4"])
def test_conditional_unet_gradients_match_autograd(backend, cfg, B, S):
    """`model(noisy, t, batch["encoding"])`, mse_loss, backward (scripts/train_unet.py:254-259): loss and every parameter
    gradient of the native forward+backward vs torch autograd on the oracle; then one optimizer step."""
    import torch.nn.functional as F
    dev = select(backend)
    from audiodiffusion import training as T
    from audiodiffusion.unet import UNet2DConditionModel
    torch.manual_seed(0)
    ref = OracleCond(**cfg)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "norm" in n:
                p.add_(0.2 * torch.randn_like(p))
    mine = UNet2DConditionModel(**cfg).load_state_dict(ref.state_dict())
    flat, grads = mine.enable_training()
    ss = cfg["sample_size"]
    hw = (ss, ss) if isinstance(ss, int) else ss
    g = torch.Generator().manual_seed(1)
    x = torch.randn((B, cfg["in_channels"]) + tuple(hw), generator=g)
    tgt = torch.randn((B, cfg["out_channels"]) + tuple(hw), generator=g)
    enc = torch.randn((B, S, cfg["cross_attention_dim"]), generator=g)
    ts = torch.tensor([5, 500, 999][:B])
    loss_ref = F.mse_loss(ref(x, ts, enc)["sample"], tgt)
    loss_ref.backward()
    loss = mine.train_step(x.to(dev), ts, tgt.to(dev), enc.to(dev))
    assert abs(float(loss) - float(loss_ref.detach())) <= 1e-5 * float(loss_ref.detach())
    gmax = max(float(p.grad.abs().max()) for p in ref.parameters())
    worst = ("", 0.0)
    for name, p in ref.named_parameters():
        off = mine.flat.offsets[name][0]
        got = grads[off:off + p.numel()].view(p.shape).cpu()
        err = float((got - p.grad).abs().max()) / max(float(p.grad.abs().max()), 1e-3 * gmax)
        if err > worst[1]:
            worst = (name, err)
    assert worst[1] < 3e-4, worst
    opt = T.AdamW(flat, lr=1e-4)
    opt.step(grads, clip=T.clip_grad_norm_(grads, 1.0))
    mine.refresh_weights()
    # data-parallel overlap: every bucket of the flat gradient buffer is reported ready exactly once by the reverse pass
    # (LayerNorm affine, to_k / to_v, bias-free projections included), so no bucket waits for the end of the backward
    import ctypes as C
    from audiodiffusion import _native as N
    per = 4096
    lows = list(range(0, grads.numel(), per)) + [grads.numel()]
    fired = []
    cb = N.BUCKET_FN(lambda _u, b: fired.append(b))
    N.check(N.lib().adm_unet_set_grad_bucket_hook(mine._handle, len(lows) - 1, (C.c_long * len(lows))(*lows),
                                                  C.cast(cb, C.c_void_p), None))
    loss2 = float(mine.train_step(x.to(dev), ts, tgt.to(dev), enc.to(dev)))
    N.check(N.lib().adm_unet_set_grad_bucket_hook(mine._handle, 0, None, None, None))
    assert sorted(fired) == list(range(len(lows) - 1))
    assert loss2 < float(loss)
