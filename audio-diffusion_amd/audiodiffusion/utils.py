"""CompVis latent-diffusion VAE checkpoint -> diffusers `AutoencoderKL` directory (reference: `audiodiffusion/utils.py:132-303`,
used by `scripts/train_vae.py:128-177` after every checkpoint to publish the VAE in the layout the pipeline loads).

Same entry points and argument meaning as the reference (`create_vae_diffusers_config`, `convert_ldm_vae_checkpoint`,
`convert_ldm_to_hf_vae`); the key translation is written as one renaming rule set (regular expressions) instead of the
reference's path-list bookkeeping, and emits the attention names diffusers 0.24 saves (`to_q/to_k/to_v/to_out.0`; the
reference emits `query/key/value/proj_attn`, which the loader maps to the same parameters).  `ldm_config` may be the
OmegaConf object the reference passes, a nested dict, or a path to the YAML file (`config/ldm_autoencoder_kl.yaml`).
"""
import re

import torch

from .vae import AutoencoderKL


def _get(node, name):
    return node[name] if isinstance(node, dict) else getattr(node, name)


def create_vae_diffusers_config(original_config):
    """diffusers AutoencoderKL kwargs from the LDM `model.params.ddconfig` block (utils.py:132-153)."""
    if isinstance(original_config, str):
        import yaml
        with open(original_config) as f:
            original_config = yaml.safe_load(f)
    params = _get(_get(original_config, "model"), "params")
    dd = _get(params, "ddconfig")
    _get(params, "embed_dim")                       # must exist (the reference reads it too)
    widths = [int(_get(dd, "ch")) * int(m) for m in _get(dd, "ch_mult")]
    res = _get(dd, "resolution")
    return dict(
        sample_size=tuple(res) if isinstance(res, (list, tuple)) else (int(res), int(res)),
        in_channels=int(_get(dd, "in_channels")),
        out_channels=int(_get(dd, "out_ch")),
        down_block_types=("DownEncoderBlock2D",) * len(widths),
        up_block_types=("UpDecoderBlock2D",) * len(widths),
        block_out_channels=tuple(widths),
        latent_channels=int(_get(dd, "z_channels")),
        layers_per_block=int(_get(dd, "num_res_blocks")),
    )


_ATTN = {"norm": "group_norm", "q": "to_q", "k": "to_k", "v": "to_v", "proj_out": "to_out.0"}


def _translate(key, n_up):
    """LDM parameter name -> diffusers name (None: not a VAE weight, e.g. the `loss.*` discriminator)."""
    if re.fullmatch(r"(encoder|decoder)\.(conv_in|conv_out)\.(weight|bias)|(post_)?quant_conv\.(weight|bias)", key):
        return key
    m = re.fullmatch(r"(encoder|decoder)\.norm_out\.(weight|bias)", key)
    if m:
        return f"{m[1]}.conv_norm_out.{m[2]}"
    m = re.fullmatch(r"encoder\.down\.(\d+)\.block\.(\d+)\.(.+)", key)
    if m:
        return f"encoder.down_blocks.{m[1]}.resnets.{m[2]}.{m[3].replace('nin_shortcut', 'conv_shortcut')}"
    m = re.fullmatch(r"encoder\.down\.(\d+)\.downsample\.conv\.(weight|bias)", key)
    if m:
        return f"encoder.down_blocks.{m[1]}.downsamplers.0.conv.{m[2]}"
    m = re.fullmatch(r"decoder\.up\.(\d+)\.block\.(\d+)\.(.+)", key)        # LDM counts up-levels from the output side
    if m:
        return f"decoder.up_blocks.{n_up - 1 - int(m[1])}.resnets.{m[2]}.{m[3].replace('nin_shortcut', 'conv_shortcut')}"
    m = re.fullmatch(r"decoder\.up\.(\d+)\.upsample\.conv\.(weight|bias)", key)
    if m:
        return f"decoder.up_blocks.{n_up - 1 - int(m[1])}.upsamplers.0.conv.{m[2]}"
    m = re.fullmatch(r"(encoder|decoder)\.mid\.block_(\d+)\.(.+)", key)
    if m:
        return f"{m[1]}.mid_block.resnets.{int(m[2]) - 1}.{m[3].replace('nin_shortcut', 'conv_shortcut')}"
    m = re.fullmatch(r"(encoder|decoder)\.mid\.attn_1\.(norm|q|k|v|proj_out)\.(weight|bias)", key)
    if m:
        return f"{m[1]}.mid_block.attentions.0.{_ATTN[m[2]]}.{m[3]}"
    return None


def convert_ldm_vae_checkpoint(checkpoint, config=None):
    """LDM VAE state dict -> diffusers AutoencoderKL state dict (utils.py:156-291).  The 1x1-convolution attention
    projections become Linear weights; keys that are not VAE weights (loss / discriminator) are dropped."""
    n_up = len({k.split(".")[2] for k in checkpoint if k.startswith("decoder.up.")})
    out = {}
    for k, v in checkpoint.items():
        nk = _translate(k, n_up)
        if nk is None:
            continue
        if ".attentions." in nk and nk.endswith(".weight") and v.ndim > 2:
            v = v.reshape(v.shape[0], v.shape[1])
        out[nk] = v
    return out


def convert_ldm_to_hf_vae(ldm_checkpoint, ldm_config, hf_checkpoint, sample_size=None):
    """utils.py:294-303: load the Lightning checkpoint, convert, save a diffusers AutoencoderKL directory."""
    try:
        ckpt = torch.load(ldm_checkpoint, map_location="cpu", weights_only=True)
    except Exception:      # Lightning checkpoints carry optimizer / callback objects next to "state_dict"
        ckpt = torch.load(ldm_checkpoint, map_location="cpu", weights_only=False)
    vae_config = create_vae_diffusers_config(ldm_config)
    if sample_size is not None:
        vae_config["sample_size"] = tuple(sample_size) if isinstance(sample_size, (list, tuple)) else (sample_size, sample_size)
    vae = AutoencoderKL(**vae_config)
    vae.load_state_dict(convert_ldm_vae_checkpoint(ckpt["state_dict"], vae_config))
    vae.save_pretrained(hf_checkpoint)
    return vae
