"""Oracle restatement of the reference's AudioEncoder forward pass (TEST INFRASTRUCTURE ONLY).

Follows `audiodiffusion/audio_encoder.py:7-84` in eval mode (`encode`, `:86-88`, calls `self.eval()` under `no_grad`):
three ConvBlocks (`:25-41`: depthwise 3x3 without bias + pointwise 1x1 `:7-22`, LeakyReLU(0.2), BatchNorm2d(eps=1e-3) on the
running statistics, MaxPool 2x2, Dropout = identity), the DenseBlock (`:44-59`: flatten of x.permute(0, 2, 3, 1), Linear,
LeakyReLU(0.2), BatchNorm1d(eps=1e-3)) and the embedding Linear (`:76,82`).  Written as plain functions over a state dict
with the reference's key names.  Parity unpinned: the reference ships no weights or golden outputs for this model.
"""
import math

import torch
import torch.nn.functional as F

CHANNELS = (1, 32, 64, 128)      # audio_encoder.py:74
EMBED = (1024, 100)              # :75-76
BN_EPS = 1e-3                    # :30,49


def forward(sd, x):
    """x (n, 1, y_res, x_res) in [0, 1] (`:92-99`: uint8 image / 255) -> (n, 100)."""
    for i in range(3):
        p = f"conv_blocks.{i}."
        x = F.conv2d(x, sd[p + "sep_conv.depthwise.weight"], None, padding=1, groups=x.shape[1])
        x = F.conv2d(x, sd[p + "sep_conv.pointwise.weight"], sd[p + "sep_conv.pointwise.bias"])
        x = F.leaky_relu(x, 0.2)
        x = F.batch_norm(x, sd[p + "batch_norm.running_mean"], sd[p + "batch_norm.running_var"], sd[p + "batch_norm.weight"],
                         sd[p + "batch_norm.bias"], False, 0.0, BN_EPS)
        x = F.max_pool2d(x, 2)
    x = x.permute(0, 2, 3, 1).flatten(1)
    x = F.leaky_relu(F.linear(x, sd["dense_block.dense.weight"], sd["dense_block.dense.bias"]), 0.2)
    x = F.batch_norm(x, sd["dense_block.batch_norm.running_mean"], sd["dense_block.batch_norm.running_var"],
                     sd["dense_block.batch_norm.weight"], sd["dense_block.batch_norm.bias"], False, 0.0, BN_EPS)
    return F.linear(x, sd["embedding.weight"], sd["embedding.bias"])


def random_state_dict(seed=0, y_res=96, x_res=216):
    """torch-default initialisation of every layer plus non-trivial BatchNorm statistics (so that the fold is exercised)."""
    g = torch.Generator().manual_seed(seed)

    def u(shape, fan_in):
        return (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)

    def bn(p, c, sd):
        sd[p + "weight"] = 1 + 0.3 * torch.randn(c, generator=g)
        sd[p + "bias"] = 0.2 * torch.randn(c, generator=g)
        sd[p + "running_mean"] = 0.1 * torch.randn(c, generator=g)
        sd[p + "running_var"] = 0.5 + torch.rand(c, generator=g)
        sd[p + "num_batches_tracked"] = torch.tensor(100)

    sd = {}
    for i in range(3):
        ci, co = CHANNELS[i], CHANNELS[i + 1]
        p = f"conv_blocks.{i}."
        sd[p + "sep_conv.depthwise.weight"] = u((ci, 1, 3, 3), 9)
        sd[p + "sep_conv.pointwise.weight"] = u((co, ci, 1, 1), ci)
        sd[p + "sep_conv.pointwise.bias"] = u((co,), ci)
        bn(p + "batch_norm.", co, sd)
    feat = CHANNELS[3] * (y_res // 8) * (x_res // 8)
    sd["dense_block.dense.weight"] = u((EMBED[0], feat), feat)
    sd["dense_block.dense.bias"] = u((EMBED[0],), feat)
    bn("dense_block.batch_norm.", EMBED[0], sd)
    sd["embedding.weight"] = u((EMBED[1], EMBED[0]), EMBED[0])
    sd["embedding.bias"] = u((EMBED[1],), EMBED[0])
    return sd
