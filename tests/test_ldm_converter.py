"""LDM -> diffusers VAE converter (audiodiffusion/utils.py:132-303): a CompVis-format checkpoint written from the oracle VAE's
weights with an independently coded name mapping must convert, load into the native AutoencoderKL and reproduce the oracle."""
import re

import pytest
import torch

from native_backend import BACKENDS, select
from oracle.vae import AutoencoderKL as OracleVAE

LDM_CFG = {"model": {"params": {"embed_dim": 1, "ddconfig": {"double_z": True, "z_channels": 1, "resolution": 32,
                                                           "in_channels": 1, "out_ch": 1, "ch": 32, "ch_mult": [1, 2],
                                                           "num_res_blocks": 1, "attn_resolutions": [], "dropout": 0.0}}}}


def _to_ldm(sd, n_blocks):
    """diffusers -> CompVis names (the inverse direction, written separately from the product code)."""
    out = {}
    for k, v in sd.items():
        k2 = k.replace("conv_norm_out", "norm_out").replace("conv_shortcut", "nin_shortcut")
        k2 = re.sub(r"encoder\.down_blocks\.(\d+)\.resnets\.(\d+)\.", r"encoder.down.\1.block.\2.", k2)
        k2 = re.sub(r"encoder\.down_blocks\.(\d+)\.downsamplers\.0\.conv\.", r"encoder.down.\1.downsample.conv.", k2)
        k2 = re.sub(r"decoder\.up_blocks\.(\d+)\.resnets\.(\d+)\.",
                    lambda m: f"decoder.up.{n_blocks - 1 - int(m[1])}.block.{m[2]}.", k2)
        k2 = re.sub(r"decoder\.up_blocks\.(\d+)\.upsamplers\.0\.conv\.",
                    lambda m: f"decoder.up.{n_blocks - 1 - int(m[1])}.upsample.conv.", k2)
        k2 = re.sub(r"mid_block\.resnets\.(\d+)\.", lambda m: f"mid.block_{int(m[1]) + 1}.", k2)
        for new, old in (("group_norm", "norm"), ("to_q", "q"), ("to_k", "k"), ("to_v", "v"), ("to_out.0", "proj_out")):
            k2 = k2.replace(f"mid_block.attentions.0.{new}.", f"mid.attn_1.{old}.")
        if "mid.attn_1" in k2 and k2.endswith("weight") and v.ndim == 2:
            v = v[:, :, None, None]                    # CompVis attention projections are 1x1 convolutions
        out[k2] = v
    return out


def test_config_from_the_reference_yaml_shape():
    from audiodiffusion.utils import create_vae_diffusers_config
    cfg = create_vae_diffusers_config({"model": {"params": {"embed_dim": 1, "ddconfig": {
        "z_channels": 1, "resolution": 256, "in_channels": 1, "out_ch": 1, "ch": 128, "ch_mult": [1, 2, 4, 4],
        "num_res_blocks": 2}}}})            # config/ldm_autoencoder_kl.yaml:18-28
    assert cfg["block_out_channels"] == (128, 256, 512, 512) and cfg["layers_per_block"] == 2
    assert cfg["latent_channels"] == 1 and cfg["sample_size"] == (256, 256)
    assert cfg["down_block_types"] == ("DownEncoderBlock2D",) * 4 and cfg["up_block_types"] == ("UpDecoderBlock2D",) * 4


@pytest.mark.parametrize("backend", BACKENDS)
def test_converted_checkpoint_reproduces_the_vae(backend, tmp_path):
    dev = select(backend)
    from audiodiffusion.utils import convert_ldm_to_hf_vae, convert_ldm_vae_checkpoint, create_vae_diffusers_config
    from audiodiffusion.vae import AutoencoderKL
    torch.manual_seed(0)
    cfg = create_vae_diffusers_config(LDM_CFG)
    ref = OracleVAE(**cfg).eval()
    ldm = _to_ldm(ref.state_dict(), 2)
    assert any(k.startswith("decoder.up.1.upsample") for k in ldm) and any("mid.attn_1.q.weight" in k for k in ldm)
    ldm["loss.discriminator.main.0.weight"] = torch.zeros(4)           # Lightning checkpoints also hold the loss module
    back = convert_ldm_vae_checkpoint(dict(ldm))
    assert set(back) == set(ref.state_dict())
    assert all(torch.equal(back[k], v) for k, v in ref.state_dict().items())
    torch.save({"state_dict": ldm, "epoch": 3}, tmp_path / "last.ckpt")
    convert_ldm_to_hf_vae(str(tmp_path / "last.ckpt"), LDM_CFG, str(tmp_path / "vae"), 32)
    vae = AutoencoderKL.from_pretrained(str(tmp_path / "vae"))
    x = torch.randn(2, 1, 32, 32, generator=torch.Generator().manual_seed(1))
    noise = torch.randn(2, 1, 16, 16, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        post = ref.encode(x).latent_dist
        want = ref.decode(post.mean + post.std * noise)["sample"]
    z = vae.encode(x.to(dev)).latent_dist.sample(noise=noise.to(dev))
    got = vae.decode(z)["sample"].cpu()
    assert float((got - want).abs().max()) <= 1e-3
