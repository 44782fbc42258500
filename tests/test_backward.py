"""Backward kernels of the training step: every gradient is checked against torch autograd
of the plain fp32 op. Tolerance 1e-4 * max|ref| (single-step grad parity bar of SURVEY.md §7.1 phase 10)."""
import pytest
import torch
import torch.nn.functional as F

from native_backend import BACKENDS, select
from test_kernels import _rand, _relerr


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [
    # (N, Cin, Cout, H, W, ks, stride, up)
    (2, 32, 64, 8, 16, 3, 1, 0),
    (1, 64, 32, 16, 16, 3, 2, 0),     # stride-2 forward conv -> zero-insertion dgrad
    (2, 32, 32, 8, 8, 3, 1, 1),       # nearest-x2 folded forward conv -> dgrad at 16x16 then 2x2 sum-pool
    (2, 96, 32, 8, 16, 1, 1, 0),      # 1x1
], ids=["s1", "s2", "up", "1x1"])
def test_conv_dgrad(backend, case):
    dev = select(backend)
    from audiodiffusion import ops
    Nn, Ci, Co, H, W, ks, stride, up = case
    x = _rand((Nn, Ci, H, W), 1, "cpu").requires_grad_(True)
    w = _rand((Co, Ci, ks, ks), 2, "cpu", scale=(Ci * ks * ks) ** -0.5)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    y = F.conv2d(xin, w, None, stride=stride, padding=ks // 2)
    dy = _rand(tuple(y.shape), 3, "cpu")
    y.backward(dy)
    wT = ops.pack_conv_weight_T(w.to(dev))
    zero = torch.zeros(Ci, device=dev)
    if stride == 2:
        dx = ops.conv2d(dy.to(dev), wT, zero, ks, up=2)
    elif up:
        dxu = ops.conv2d(dy.to(dev), wT, zero, ks)
        dx = ops.sumpool2x2(dxu)
    else:
        dx = ops.conv2d(dy.to(dev), wT, zero, ks)
    assert dx.shape == x.grad.shape
    assert _relerr(dx, x.grad) < 1e-4


WG_CASES = [
    # (N, C1, C2, H, W, Cout, ks, stride, up, gn, act)
    (2, 32, 0, 8, 16, 32, 3, 1, 0, 1, 1),
    (1, 32, 32, 16, 16, 64, 3, 1, 0, 1, 1),     # virtual concat
    (2, 64, 0, 16, 16, 32, 3, 2, 0, 0, 0),      # stride 2
    (3, 32, 0, 4, 4, 32, 3, 1, 1, 0, 0),        # upsample folded
    (2, 64, 32, 8, 8, 96, 1, 1, 0, 0, 0),       # 1x1 over concat
    (2, 32, 0, 2, 2, 32, 3, 1, 0, 1, 1),        # 2x2 level
    (1, 160, 0, 8, 16, 160, 3, 1, 0, 1, 1),     # > 128 couts / channels: several tiles and chunks
    (3, 32, 64, 8, 32, 128, 3, 1, 0, 1, 1),     # full 128-cout tile, chunks inside x1 / x2: the tile-invariant prefetch path
    (2, 64, 0, 16, 16, 256, 3, 1, 0, 0, 0),     # ... without GroupNorm, two cout tiles, image borders on every side
    (2, 128, 128, 8, 8, 128, 1, 1, 0, 0, 0),    # ... 1x1 (128-channel chunks) over a concat
    (5, 32, 0, 4, 4, 128, 3, 1, 0, 1, 1),       # ... several images per 64-pixel tile, ragged last image group
    (2, 32, 0, 4, 4, 32, 3, 2, 0, 0, 0),        # stride 2 down to 2x2 (deepest Downsample2D of a 64x64 model): 16 images x 5x5
    (3, 32, 0, 2, 2, 32, 3, 2, 0, 0, 0),        # ... down to 1x1 (32x32 model): 64 images x 3x3 patches per tile, 107 KiB of LDS
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("max_split", [0, 2], ids=["split-auto", "split2"])
@pytest.mark.parametrize("case", WG_CASES, ids=[str(i) for i in range(len(WG_CASES))])
def test_conv_wgrad(backend, case, max_split):
    """max_split = 2 puts many pixel tiles on one workgroup: the software-pipelined kernels' steady state (LDS double
    buffer, loads two tiles ahead) instead of prologue/epilogue only."""
    dev = select(backend)
    from audiodiffusion import _native
    _native.check(_native.lib().adm_set_option(b"wgrad_max_split", max_split))
    try:
        _wgrad_case(dev, case)
    finally:
        _native.check(_native.lib().adm_set_option(b"wgrad_max_split", 0))


def _wgrad_case(dev, case):
    from audiodiffusion import ops
    Nn, C1, C2, H, W, Cout, ks, stride, up, use_gn, act = case
    x1 = _rand((Nn, C1, H, W), 1, dev)
    x2 = _rand((Nn, C2, H, W), 2, dev) if C2 else None
    Ct = C1 + C2
    w = _rand((Cout, Ct, ks, ks), 3, "cpu", scale=(Ct * ks * ks) ** -0.5).requires_grad_(True)
    gamma, beta = _rand((Ct,), 5, dev), _rand((Ct,), 6, dev)
    xc = torch.cat([x1, x2], 1).cpu() if C2 else x1.cpu()
    a = xc
    if use_gn:
        a = F.group_norm(a, 32, gamma.cpu(), beta.cpu(), 1e-5)
    if act:
        a = F.silu(a)
    if up:
        a = F.interpolate(a, scale_factor=2.0, mode="nearest")
    y = F.conv2d(a, w, None, stride=stride, padding=ks // 2)
    dy = _rand(tuple(y.shape), 7, "cpu")
    y.backward(dy)
    gn = ops.groupnorm_stats(x1, gamma, beta, 32, 1e-5, x2=x2) if use_gn else None
    dW = ops.conv2d_wgrad(x1, dy.to(dev), Cout, ks, x2=x2, up=bool(up), stride=stride, gn=gn, act=bool(act))
    assert dW.shape == w.grad.shape
    assert _relerr(dW, w.grad) < 1e-4, _relerr(dW, w.grad)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("C1,C2,HW,act", [(32, 0, 64, 1), (64, 32, 16, 1), (32, 0, 256, 0), (96, 96, 4, 1)])
def test_groupnorm_silu_backward(backend, C1, C2, HW, act):
    dev = select(backend)
    from audiodiffusion import ops
    h = int(round(HW ** 0.5))
    x1 = (_rand((2, C1, h, HW // h), 1, "cpu") * 2 + 0.5).requires_grad_(True)
    x2 = _rand((2, C2, h, HW // h), 2, "cpu").requires_grad_(True) if C2 else None
    gamma = (_rand((C1 + C2,), 3, "cpu") + 1.0).requires_grad_(True)
    beta = _rand((C1 + C2,), 4, "cpu").requires_grad_(True)
    xc = torch.cat([x1, x2], 1) if C2 else x1
    a = F.group_norm(xc, 32, gamma, beta, 1e-5)
    if act:
        a = F.silu(a)
    da = _rand(tuple(a.shape), 5, "cpu")
    a.backward(da)
    d = lambda t: None if t is None else t.detach().to(dev)  # noqa: E731
    _, _, mr = ops.groupnorm_stats_ex(d(x1), d(gamma), d(beta), 32, 1e-5, x2=d(x2))
    dx1, dx2, dg, db = ops.groupnorm_backward(d(x1), da.to(dev), mr, d(gamma), d(beta), 32, bool(act), x2=d(x2))
    assert _relerr(dx1, x1.grad) < 1e-4
    if C2:
        assert _relerr(dx2, x2.grad) < 1e-4
    assert _relerr(dg, gamma.grad) < 1e-4 and _relerr(db, beta.grad) < 1e-4


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("C,T,d", [(32, 64, 8), (64, 256, 8), (32, 16, 8)])
def test_attention_backward(backend, C, T, d):
    dev = select(backend)
    from audiodiffusion import ops
    h = int(round(T ** 0.5))
    qkv = _rand((2, 3 * C, h, T // h), 1, "cpu").requires_grad_(True)
    q, k, v = qkv.view(2, 3, C // d, d, T).unbind(1)
    s = torch.einsum("nhdt,nhdj->nhtj", q, k) * d ** -0.5
    out = torch.einsum("nhtj,nhdj->nhdt", s.softmax(-1), v).reshape(2, C, h, T // h)
    do = _rand(tuple(out.shape), 2, "cpu")
    out.backward(do)
    dqkv = ops.attention_backward(qkv.detach().to(dev), do.to(dev), d)
    assert _relerr(dqkv, qkv.grad) < 1e-4


@pytest.mark.parametrize("backend", BACKENDS)
def test_small_ops_backward(backend):
    dev = select(backend)
    from audiodiffusion import ops
    # chan_sums
    dy = _rand((3, 32, 8, 8), 1, dev)
    nc, c = ops.chan_sums(dy)
    assert torch.allclose(nc.cpu(), dy.cpu().sum((2, 3)), atol=1e-4) and torch.allclose(c.cpu(), dy.cpu().sum((0, 2, 3)), atol=1e-4)
    # linear backward with silu on the input (time_emb_proj) and without (time embedding MLP)
    for x_silu in (True, False):
        X = _rand((4, 64), 2, "cpu").requires_grad_(True)
        Wt = _rand((96, 64), 3, "cpu", 0.1).requires_grad_(True)
        b = _rand((96,), 4, "cpu").requires_grad_(True)
        Y = F.linear(F.silu(X) if x_silu else X, Wt, b)
        dY = _rand((4, 96), 5, "cpu")
        Y.backward(dY)
        dW, db, dX = ops.linear_backward(dY.to(dev), X.detach().to(dev), Wt.detach().to(dev), x_silu=x_silu)
        assert _relerr(dW, Wt.grad) < 1e-5 and _relerr(db, b.grad) < 1e-5 and _relerr(dX, X.grad) < 1e-5
    # conv_in class weight gradient
    x = _rand((2, 1, 16, 32), 6, "cpu")
    w = _rand((32, 1, 3, 3), 7, "cpu", 0.3).requires_grad_(True)
    y = F.conv2d(x, w, None, padding=1)
    dyc = _rand(tuple(y.shape), 8, "cpu")
    y.backward(dyc)
    assert _relerr(ops.conv_small_cin_wgrad(x.to(dev), dyc.to(dev)), w.grad) < 1e-4
    # conv_out class: GN + SiLU -> 32 -> 1
    hx = _rand((2, 32, 16, 32), 9, "cpu")
    gamma, beta = _rand((32,), 10, "cpu") + 1, _rand((32,), 11, "cpu")
    w2 = _rand((1, 32, 3, 3), 12, "cpu", 0.1).requires_grad_(True)
    a = F.silu(F.group_norm(hx, 32, gamma, beta, 1e-5)).requires_grad_(True)
    y2 = F.conv2d(a, w2, None, padding=1)
    dy2 = _rand(tuple(y2.shape), 13, "cpu")
    y2.backward(dy2)
    gn = ops.groupnorm_stats(hx.to(dev), gamma.to(dev), beta.to(dev), 32, 1e-5)
    da, dW2 = ops.conv_small_cout_backward(hx.to(dev), w2.detach().to(dev), dy2.to(dev), gn=gn, act=True)
    assert _relerr(da, a.grad) < 1e-4 and _relerr(dW2, w2.grad) < 1e-4
