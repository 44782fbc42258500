"""Seeded random sweep over the fused convolution's argument space: whichever kernel the dispatcher picks (persistent
Winograd, pipelined / plain implicit-GEMM, pointwise, small-channel direct) must agree with torch fp32. Guards the
eligibility logic between the variants (tile divisibility, concat seams, upsample fold, stride 2, missing GroupNorm)."""
import random

import pytest
import torch

from native_backend import BACKENDS, select
from test_kernels import _conv_ref, _rand, _relerr


def _cases(n, seed):
    rng = random.Random(seed)
    out = []
    while len(out) < n:
        ks = rng.choice([3, 3, 3, 1])
        stride = rng.choice([1, 1, 1, 2]) if ks == 3 else 1
        up = rng.choice([0, 0, 1]) if (ks == 3 and stride == 1) else 0
        H, W = rng.choice([2, 4, 8, 16, 24]), rng.choice([2, 4, 8, 16, 32, 48])
        if stride == 2 and (H % 2 or W % 2):
            continue
        if H < 4 and W > 8:
            continue
        C1 = rng.choice([32, 32, 64, 96])
        C2 = rng.choice([0, 0, 32, 64])
        Cout = rng.choice([32, 64, 64, 128, 192])
        gn = rng.choice([0, 1, 1])
        act = rng.choice([0, 1]) if gn else 0
        if up and (C2 or gn):
            gn, act = 0, 0              # Upsample2D.conv has no norm on its input; keep concat + upsample out (not in the path)
            C2 = 0
        # output sizes below 16x8 must be powers of two for the implicit-GEMM tiling
        Ho = (2 * H if up else H) // stride
        Wo = (2 * W if up else W) // stride
        if (Ho < 8 and Ho & (Ho - 1)) or (Wo < 16 and Wo & (Wo - 1)):
            continue
        out.append((rng.choice([1, 2, 3]), C1, C2, H, W, Cout, ks, stride, up, gn, act, rng.choice([0, 1]), rng.choice([0, 1])))
    return out


CASES = _cases(40, seed=20260924)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", CASES, ids=["-".join(map(str, c)) for c in CASES])
def test_conv2d_random_dispatch(backend, case):
    dev = select(backend)
    from audiodiffusion import _native, ops
    Nn, C1, C2, H, W, Cout, ks, stride, up, use_gn, act, use_temb, use_res = case
    x1 = _rand((Nn, C1, H, W), 1, dev)
    x2 = _rand((Nn, C2, H, W), 2, dev) if C2 else None
    Ct = C1 + C2
    w = _rand((Cout, Ct, ks, ks), 3, dev, scale=(Ct * ks * ks) ** -0.5)
    b = _rand((Cout,), 4, dev)
    gamma, beta = _rand((Ct,), 5, dev), _rand((Ct,), 6, dev)
    gn = ops.groupnorm_stats(x1, gamma, beta, 32, 1e-5, x2=x2) if use_gn else None
    temb = _rand((Nn, Cout), 7, dev) if use_temb else None
    Hi, Wi = (2 * H, 2 * W) if up else (H, W)
    Ho, Wo = (Hi, Wi) if stride == 1 else (Hi // 2, Wi // 2)
    res = _rand((Nn, Cout, Ho, Wo), 8, dev) if use_res else None
    wino = ops.pack_winograd_weight(w) if (ks == 3 and stride == 1) else None       # as the network executor does
    out = ops.conv2d(x1, ops.pack_conv_weight(w), b, ks, x2=x2, up=bool(up), stride=stride, pad_lo=1, gn=gn,
                     act=bool(act), chan_add=temb, residual=res, wino=wino)
    variant = _native.lib().adm_last_conv_variant()
    c = lambda t: None if t is None else t.cpu()  # noqa: E731
    ref = _conv_ref(c(x1), c(x2), c(w), c(b), ks, stride, up, (c(gamma), c(beta)) if use_gn else None, act, c(temb), c(res))
    assert out.shape == ref.shape
    assert _relerr(out, ref) < 1e-4, (variant, _relerr(out, ref))
    # the persistent Winograd kernel must be the one that ran whenever the shape allows it
    eligible = (ks == 3 and stride == 1 and Wi % 16 == 0 and Hi % 8 == 0 and Ct % 16 == 0 and C1 % 8 == 0 and Cout % 64 == 0
                and (use_gn or not act))
    assert (variant in (4313, 4314, 4315, 4316)) == eligible, (variant, eligible)
    if eligible:      # filters L2 -> registers (v4 / v5 / v6) whenever 32 | input channels, else the LDS-DMA kernel (v3)
        if Ct % 32 != 0:
            assert variant == 4313, (variant, Ct)
        elif (Cout % 128 == 0 and C1 % 16 == 0 and Hi % 16 == 0 and Wi % 16 == 0 and min(Hi, Wi) >= 64
              and (Hi // 16) * (Wi // 16) * (Cout // 128) >= 32):
            assert variant == 4316, (variant, Hi, Wi)          # F(4x4,3x3): by the layer alone
        elif Cout % 128 != 0:
            assert variant == 4314, (variant, Cout)
        else:         # v5 when its 128-cout tiles fill the chip (the emulator's 3 "CUs"), else v4: bit-identical either way
            tiles5 = (Wi // 16) * (Hi // 8) * Nn * (Cout // 128)
            assert variant == (4315 if tiles5 >= (3 if backend == "emu" else 256) else 4314), (variant, tiles5)


def _wg_cases(n, seed):
    rng = random.Random(seed)
    out = []
    while len(out) < n:
        ks = rng.choice([3, 3, 1])
        stride = rng.choice([1, 1, 2]) if ks == 3 else 1
        up = rng.choice([0, 0, 1]) if (ks == 3 and stride == 1) else 0
        H, W = rng.choice([2, 4, 8, 16]), rng.choice([2, 4, 8, 16, 32])
        if H < 4 and W > 8:
            continue
        if stride == 2 and (H < 8 or W < 8):
            continue                    # the network's stride-2 convs sit at >= 16x16 inputs; tiny ones exceed the LDS patch budget (loud error)
        C1 = rng.choice([32, 32, 64, 128])
        C2 = rng.choice([0, 0, 32, 128])
        Cout = rng.choice([32, 96, 128, 128, 256])
        gn = rng.choice([0, 1])
        act = rng.choice([0, 1]) if gn else 0
        if up:
            C2, gn, act = 0, 0, 0
        out.append((rng.choice([1, 2, 5]), C1, C2, H, W, Cout, ks, stride, up, gn, act, rng.choice([0, 0, 2, 3])))
    return out


WG_RANDOM = _wg_cases(24, seed=7)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", WG_RANDOM, ids=["-".join(map(str, c)) for c in WG_RANDOM])
def test_conv_wgrad_random_dispatch(backend, case):
    """Weight gradient through whichever variant is picked (8-wave software-pipelined, pipelined, plain; tile-invariant or
    generic prefetch; any split-K factor) vs torch autograd."""
    dev = select(backend)
    from audiodiffusion import _native
    from test_backward import _wgrad_case
    _native.check(_native.lib().adm_set_option(b"wgrad_max_split", case[-1]))
    try:
        _wgrad_case(dev, case[:-1])
    finally:
        _native.check(_native.lib().adm_set_option(b"wgrad_max_split", 0))
