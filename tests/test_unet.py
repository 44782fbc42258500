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


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("cfg,B", [(TINY, 2), (TINY3, 3)], ids=["tiny2", "tiny3"])
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
    ts = torch.tensor([5, 500, 999][:B])
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
