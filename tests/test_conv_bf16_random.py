"""Seeded random sweep over the shapes the bf16 3x3 kernels accept (forward and weight gradient), each case against the float64 convolution of the same bf16-rounded operands."""
import random

import pytest
import torch
import torch.nn.functional as F

from native_backend import BACKENDS, select
from test_kernels import _rand, _relerr


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float64)


def _cases(n, seed):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        c2 = rng.choice([0, 0, 32])
        c1 = rng.choice([32, 64, 96]) if c2 == 0 else rng.choice([32, 96])
        up = rng.random() < 0.25
        h, w = rng.choice([8, 16]) if up else rng.choice([16, 32, 48]), rng.choice([8, 16]) if up else rng.choice([16, 32])
        gn = rng.random() < 0.7
        out.append((rng.randint(1, 3), c1, c2, h, w, rng.choice([128, 128, 256]), int(up), int(gn),
                    int(gn and rng.random() < 0.7), int(rng.random() < 0.5), int(rng.random() < 0.5),
                    rng.choice(["one-tile", "persist"])))
    return out


CASES = _cases(18, 20260924)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", CASES, ids=[f"{i}-{c[-1]}" for i, c in enumerate(CASES)])
def test_random_bf16_forward(backend, case):
    dev = select(backend)
    from audiodiffusion import _native, ops
    Nn, C1, C2, H, W, Cout, up, use_gn, act, use_temb, use_res, variant = case
    x1 = _rand((Nn, C1, H, W), 1, dev)
    x2 = _rand((Nn, C2, H, W), 2, dev) if C2 else None
    Ct = C1 + C2
    w = _rand((Cout, Ct, 3, 3), 3, dev, scale=(Ct * 9) ** -0.5)
    b = _rand((Cout,), 4, dev)
    gamma, beta = _rand((Ct,), 5, dev), _rand((Ct,), 6, dev)
    gn = ops.groupnorm_stats(x1, gamma, beta, 32, 1e-5, x2=x2) if use_gn else None
    temb = _rand((Nn, Cout), 7, dev) if use_temb else None
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    res = _rand((Nn, Cout, Ho, Wo), 8, dev) if use_res else None
    _native.check(_native.lib().adm_set_option(b"conv_bf16", 1))
    try:
        out = ops.conv2d(x1, ops.pack_conv_weight(w), b, 3, x2=x2, up=bool(up), gn=gn, act=bool(act), chan_add=temb,
                         residual=res, bf16=ops.pack_bf16_weight(w))
        assert _native.lib().adm_last_conv_variant() == 5316
    finally:
        _native.check(_native.lib().adm_set_option(b"conv_bf16", 0))
    c = lambda t: None if t is None else t.cpu()  # noqa: E731
    x = torch.cat([c(x1), c(x2)], 1) if C2 else c(x1)
    if use_gn:
        x = F.group_norm(x, 32, c(gamma), c(beta), 1e-5)
    if act:
        x = F.silu(x)
    if up:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(_bf(x), _bf(c(w)), None, padding=1) + c(b).double()[None, :, None, None]
    if use_temb:
        ref = ref + c(temb).double()[:, :, None, None]
    if use_res:
        ref = ref + c(res).double()
    assert _relerr(out.double(), ref) < (2e-6 if not (use_gn or act) else 3e-4)


WG_CASES = [(rng_n, c1, c2, h, w, co, up, gn, act, ms) for (rng_n, c1, c2, h, w, co, up, gn, act, _t, _r, _v), ms in
            zip(_cases(8, 7), [0, 1, 2, 0, 3, 0, 2, 1])]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", WG_CASES, ids=[str(i) for i in range(len(WG_CASES))])
def test_random_bf16_weight_gradient(backend, case):
    dev = select(backend)
    from audiodiffusion import _native, ops
    Nn, C1, C2, H, W, Cout, up, use_gn, act, max_split = case
    x1 = _rand((Nn, C1, H, W), 1, dev)
    x2 = _rand((Nn, C2, H, W), 2, dev) if C2 else None
    Ct = C1 + C2
    gamma, beta = _rand((Ct,), 5, dev), _rand((Ct,), 6, dev)
    a = torch.cat([x1, x2], 1).cpu() if C2 else x1.cpu()
    if use_gn:
        a = F.group_norm(a, 32, gamma.cpu(), beta.cpu(), 1e-5)
    if act:
        a = F.silu(a)
    if up:
        a = F.interpolate(a, scale_factor=2.0, mode="nearest")
    dy = _rand((Nn, Cout) + tuple(a.shape[2:]), 7, "cpu")
    ref = torch.nn.grad.conv2d_weight(_bf(a), (Cout, Ct, 3, 3), _bf(dy), padding=1)
    gn = ops.groupnorm_stats(x1, gamma, beta, 32, 1e-5, x2=x2) if use_gn else None
    _native.check(_native.lib().adm_set_option(b"conv_bf16", 1))
    _native.check(_native.lib().adm_set_option(b"wgrad_max_split", max_split))
    try:
        dW = ops.conv2d_wgrad(x1, dy.to(dev), Cout, 3, x2=x2, up=bool(up), gn=gn, act=bool(act))
    finally:
        _native.check(_native.lib().adm_set_option(b"conv_bf16", 0))
        _native.check(_native.lib().adm_set_option(b"wgrad_max_split", 0))
    assert _relerr(dW.double(), ref) < (2e-6 if not (use_gn or act) else 3e-4)
