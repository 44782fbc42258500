"""Whole-UNet parity: native executor (C-ABI adm_unet_forward) vs the oracle UNet2DModel with identical weights.
Tolerance: max|d eps| <= 1e-3; observed values are ~1e-5."""
import pytest
import torch

from native_backend import BACKENDS, select
from oracle.unet import UNet2DModel as OracleUNet

TINY = dict(sample_size=16, in_channels=1, out_channels=1, layers_per_block=2, block_out_channels=(32, 64),
            down_block_types=("DownBlock2D", "AttnDownBlock2D"), up_block_types=("AttnUpBlock2D", "UpBlock2D"))
TINY3 = dict(sample_size=(8, 16), in_channels=1, out_channels=1, layers_per_block=1, block_out_channels=(32, 32, 64),
             down_block_types=("DownBlock2D", "AttnDownBlock2D", "DownBlock2D"),
             up_block_types=("UpBlock2D", "AttnUpBlock2D", "UpBlock2D"))


def _pair(cfg, seed=0):
    from audiodiffusion.unet import UNet2DModel, param_specs
    torch.manual_seed(seed)
    ref = OracleUNet(**cfg).eval()
    # make GroupNorm affine / biases non-trivial so a swapped gamma/beta or missing bias shows
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "norm" in n:
                p.add_(0.2 * torch.randn_like(p))
    mine = UNet2DModel(**cfg)
    sd = ref.state_dict()
    assert {k for k, _, _ in param_specs(mine.config)} == set(sd.keys())
    mine.load_state_dict(sd)
    return ref, mine


WIDE = dict(sample_size=8, in_channels=1, out_channels=1, layers_per_block=1, block_out_channels=(128, 128),
            down_block_types=("DownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "UpBlock2D"))   # the shipped models' 128 -> 512 -> 512 time embedding


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("cfg,B", [(TINY, 2), (TINY3, 3), (WIDE, 2), (WIDE, 9)], ids=["tiny2", "tiny3", "wide", "wide9"])
def test_unet_forward_matches_oracle(backend, cfg, B):
    dev = select(backend)
    ref, mine = _pair(cfg)
    ss = cfg["sample_size"]
    hw = (ss, ss) if isinstance(ss, int) else ss
    g = torch.Generator().manual_seed(1)
    x = torch.randn((B, 1) + tuple(hw), generator=g)
    for t in (torch.tensor(980), torch.tensor(0), 37):
        with torch.no_grad():
            r = ref(x, t)["sample"]
        o = mine(x.to(dev), t)["sample"].cpu()
        assert o.shape == r.shape
        assert float((o - r).abs().max()) < 1e-3 * max(1.0, float(r.abs().max())), float((o - r).abs().max())
    # per-sample timesteps (training form, train_unet.py:241-257)
    ts = torch.tensor(([5, 500, 999] * 3)[:B])
    with torch.no_grad():
        r = ref(x, ts)["sample"]
    o = mine(x.to(dev), ts)["sample"].cpu()
    assert float((o - r).abs().max()) < 1e-3 * max(1.0, float(r.abs().max()))


def test_param_specs_count_matches_reference_config():
    from audiodiffusion.unet import UNet2DModel
    m = UNet2DModel(sample_size=256, in_channels=1, out_channels=1, layers_per_block=2,
                    block_out_channels=(128, 128, 256, 256, 512, 512),
                    down_block_types=("DownBlock2D",) * 4 + ("AttnDownBlock2D", "DownBlock2D"),
                    up_block_types=("UpBlock2D", "AttnUpBlock2D") + ("UpBlock2D",) * 4)
    assert m.num_parameters() == 113_668_609


def test_deprecated_attention_names_and_unknown_blocks():
    from audiodiffusion.unet import UNet2DModel
    select("emu")
    ref, mine = _pair(TINY)
    old = {}
    for k, v in ref.state_dict().items():
        for new, o in ((".to_q.", ".query."), (".to_k.", ".key."), (".to_v.", ".value."), (".to_out.0.", ".proj_attn.")):
            if ".attentions." in k:
                k = k.replace(new, o)
        old[k] = v
    assert any(".query." in k for k in old)
    mine.load_state_dict(old)
    with pytest.raises(NotImplementedError):
        UNet2DModel(down_block_types=("CrossAttnDownBlock2D",), up_block_types=("UpBlock2D",), block_out_channels=(32,))


SMALL_PLANES = dict(sample_size=8, in_channels=1, out_channels=1, layers_per_block=1, block_out_channels=(64, 128),
                    down_block_types=("DownBlock2D", "AttnDownBlock2D"), up_block_types=("AttnUpBlock2D", "UpBlock2D"))


ONE_PIXEL = dict(sample_size=4, in_channels=1, out_channels=1, layers_per_block=1, block_out_channels=(64, 128, 128),
                 down_block_types=("DownBlock2D", "AttnDownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "AttnUpBlock2D", "UpBlock2D"))


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("cfg", [SMALL_PLANES, ONE_PIXEL], ids=["8x8-4x4", "4x4-2x2-1x1"])
def test_groupnorm_from_the_split_k_finish_pass(backend, cfg):
    """Option gn_fuse_finish (default 1): on planes of <= 8x8 pixels a split-K convolution's finish pass leaves the scale / shift
    of the GroupNorm that reads its output (one launch instead of two). The statistics ops it replaces show up as (kind 0,
    variant 2) zero-cost records in the op profile; the forward stays inside the oracle tolerance and agrees with the two-launch path
    (the tensors are bit-identical, the statistics sum the same values in another fp64 order)."""
    import ctypes as C
    from audiodiffusion import _native as N
    dev = select(backend)
    ref, mine = _pair(cfg, seed=3)
    mine = mine.to(dev)
    B = 3
    ss = cfg["sample_size"]
    x = torch.randn(B, 1, ss, ss, generator=torch.Generator().manual_seed(2))     # (1- and 2-pixel planes: the scalar slab loop)
    with torch.no_grad():
        r = ref(x, 37)["sample"]
    outs, fused = {}, {}
    try:
        for on in (0, 1):
            N.check(N.lib().adm_set_option(b"gn_fuse_finish", on))
            o = mine(x.to(dev), 37)["sample"]
            xd = x.to(dev)
            out = torch.empty_like(xd)
            recs = (N.OpProfile * 512)()
            n = C.c_int(0)
            N.check(N.lib().adm_unet_profile(mine._ensure_handle(), N.ptr(xd), 37.0, N.ptr(out), B, recs, 512, C.byref(n), N.stream_for(xd)))
            fused[on] = sum(1 for q in recs[: n.value] if q.kind == 0 and q.variant == 2)
            assert torch.equal(out, o)
            outs[on] = o.cpu()
    finally:
        N.check(N.lib().adm_set_option(b"gn_fuse_finish", -1))
    assert fused[0] == 0 and fused[1] >= 4, fused
    for o in outs.values():
        assert float((o - r).abs().max()) < 1e-3 * max(1.0, float(r.abs().max()))
    assert float((outs[0] - outs[1]).abs().max()) < 1e-5 * max(1.0, float(r.abs().max()))


W6NET = dict(sample_size=16, in_channels=1, out_channels=1, layers_per_block=1, block_out_channels=(128, 128),
             down_block_types=("DownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "UpBlock2D"))


@pytest.mark.parametrize("backend", BACKENDS)
def test_an_option_set_between_two_forwards_re_plans_the_net(backend):
    """`set_option("wino6", ...)` moves layers between the F(4x4) and the F(2x2) kernel, whose GroupNorm partial-sum tiles differ
    (16x16 / 8x16 pixels); a net planned before the change kept the old tile counts and failed (`stats_tiles does not match`) — and a captured
    loop kept the old kernels. Forward, change the option, forward, change it back, forward: every pass matches the oracle, and the loop too."""
    import audiodiffusion
    from audiodiffusion import AudioDiffusionPipeline, DDIMScheduler, Mel, _native
    dev = select(backend)
    ref, mine = _pair(W6NET)
    x = torch.randn(2, 1, 16, 16, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        r = ref(x, 500)["sample"]
    variants = []
    try:
        for v in (2, 0, -1):                       # every layer the kernel tiles -> F(2x2) only -> the environment's default
            audiodiffusion.set_option("wino6", v)
            o = mine(x.to(dev), 500)["sample"].cpu()
            assert float((o - r).abs().max()) < 1e-3 * max(1.0, float(r.abs().max())), v
            variants.append(_native.lib().adm_last_conv_variant())
        pipe = AudioDiffusionPipeline(None, mine, Mel(x_res=16, y_res=16, hop_length=64, n_fft=256, n_iter=1), DDIMScheduler()).to(dev)
        pipe.set_progress_bar_config(disable=True)
        outs = []
        for v in (2, 0, 2):                        # the captured loop: same noise, the option changed between samplings
            audiodiffusion.set_option("wino6", v)
            outs.append(pipe(batch_size=2, steps=2, noise=x.clone().to(dev), audio=False, return_float=True)[1].cpu())
        assert torch.equal(outs[0], outs[2])       # back under the first setting: the same kernels, the same bits
        assert float((outs[0] - outs[1]).abs().max()) < 1e-3
    finally:
        audiodiffusion.set_option("wino6", -1)


@pytest.mark.parametrize("backend", BACKENDS)
def test_the_f4x4_layer_rule_is_a_per_model_setting(backend):
    """`UNet2DModel.set_option("wino6", rule)` (adm_unet_set_option): one model runs its own rule — here F(4x4) on every layer the kernel
    tiles — while a second model in the same process keeps the process-wide default (F(2x2) on these 16x16 planes); both match the oracle,
    each model's batch rows stay bit-identical to its single-sample runs, and rule 0 hands the model back to the process-wide option.
    (16x16 planes: one F(4x4) tile per sample and cout block — the kernel's smallest case.)"""
    from audiodiffusion import _native
    dev = select(backend)
    ref, mine = _pair(W6NET)
    _, other = _pair(W6NET)
    x = torch.randn(3, 1, 16, 16, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        r = ref(x, 500)["sample"]
    import ctypes as C

    def variants(m):                                               # the kernel variant of every launch of one eager forward
        recs, n = (_native.OpProfile * 512)(), C.c_int(0)
        xd, out = x.to(dev), torch.empty_like(x.to(dev))
        _native.check(_native.lib().adm_unet_profile(m._ensure_handle(), _native.ptr(xd), 500.0, _native.ptr(out), x.shape[0], recs, 512,
                                                     C.byref(n), _native.stream_for(xd)))
        return {r.variant for r in recs[: n.value]}
    mine.set_option("wino6", 2)
    a = mine(x.to(dev), 500)["sample"].cpu()
    b = other(x.to(dev), 500)["sample"].cpu()
    va, vb = variants(mine), variants(other)
    assert 4316 in va and 4316 not in vb and (vb & {4314, 4315}), (va, vb)
    for o in (a, b):
        assert float((o - r).abs().max()) < 1e-3 * max(1.0, float(r.abs().max()))
    assert not torch.equal(a, b)                                   # two roundings of the same network
    assert torch.equal(mine(x[1:2].to(dev), 500)["sample"].cpu(), a[1:2])      # a row alone = the row in its batch, under the model's rule
    mine.set_option("wino6", 0)
    assert torch.equal(mine(x.to(dev), 500)["sample"].cpu(), b)


@pytest.mark.parametrize("backend", BACKENDS)
def test_the_split_k_rule_is_a_per_model_setting(backend):
    """`UNet2DModel.set_option("single_sample", 1)` — the single-sample rule of the 64-cout F(2x2) kernel (include/adm.h): this model's 3x3
    layers split their input channels over several workgroups per tile (variant 4317) and its statistics come from the finish pass; a second
    model in the process keeps the unsplit kernels. Both match the oracle; the partition is a function of the layer, so the model's batch
    rows stay bit-identical to its single-sample runs; the captured loop runs under the rule as well; 0 hands the model back."""
    import ctypes as C
    from audiodiffusion import _native
    dev = select(backend)
    ref, mine = _pair(W6NET)
    _, other = _pair(W6NET)
    x = torch.randn(3, 1, 16, 16, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        r = ref(x, 500)["sample"]

    def variants(m):
        recs, n = (_native.OpProfile * 512)(), C.c_int(0)
        xd, out = x.to(dev), torch.empty_like(x.to(dev))
        _native.check(_native.lib().adm_unet_profile(m._ensure_handle(), _native.ptr(xd), 500.0, _native.ptr(out), x.shape[0], recs, 512,
                                                     C.byref(n), _native.stream_for(xd)))
        return {r.variant for r in recs[: n.value]}
    mine.set_option("single_sample", 1)
    a = mine(x.to(dev), 500)["sample"].cpu()
    b = other(x.to(dev), 500)["sample"].cpu()
    va, vb = variants(mine), variants(other)
    assert 4317 in va and 4317 not in vb and not (va & {4314, 4315}) and (vb & {4314, 4315}), (va, vb)
    for o in (a, b):
        assert float((o - r).abs().max()) < 1e-3 * max(1.0, float(r.abs().max()))
    assert not torch.equal(a, b)                                   # two summation orders of the same network
    for i in range(3):
        assert torch.equal(mine(x[i:i + 1].to(dev), 500)["sample"].cpu(), a[i:i + 1])
    with pytest.raises(RuntimeError, match="single_sample"):
        mine.set_option("single_sample", 7)
    assert mine._options == {"single_sample": 1}                     # the rejected value was not remembered
    mine.set_option("single_sample", 0)
    assert torch.equal(mine(x.to(dev), 500)["sample"].cpu(), b)
