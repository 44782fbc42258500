"""Round 4: the 16-bit training convolutions on BLOCKED operand images (k_conv_bf16b.hip; option conv_bf16 = 3).
`--mixed_precision bf16` (scripts/train_unet.py:391-401 -> accelerate -> torch.autocast: Conv2d on bf16 operands, fp32 accumulation).
Bars as in test_conv_bf16.py: (tight) against a float64 convolution of the SAME bf16-rounded operands — only the accumulation
order differs; (loose) against the plain fp32 convolution — the error of the mixed-precision mode itself."""
import pytest
import torch
import torch.nn.functional as F

from native_backend import BACKENDS, select
from test_kernels import _rand, _relerr


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float64)


def _unblock(img):
    """(N, C/8, H+2, W+2, 8) -> (N, C, H+2, W+2) float64"""
    n, cg, hp, wp, _ = img.shape
    return img.permute(0, 1, 4, 2, 3).reshape(n, cg * 8, hp, wp).to(torch.float64)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [(2, 16, 8, 8, 12, True, True), (1, 8, 0, 5, 8, False, False), (1, 32, 32, 16, 64, True, False)],
                         ids=["concat-gn-silu", "bare", "wide"])
def test_blocked_image_is_the_rounded_activated_input_with_a_zero_halo(backend, case):
    dev = select(backend)
    from audiodiffusion import ops
    Nn, C1, C2, H, W, use_gn, act = case
    x1 = _rand((Nn, C1, H, W), 1, dev)
    x2 = _rand((Nn, C2, H, W), 2, dev) if C2 else None
    Ct = C1 + C2
    gamma, beta = _rand((Ct,), 5, dev), _rand((Ct,), 6, dev)
    gn = ops.groupnorm_stats(x1, gamma, beta, 8, 1e-5, x2=x2) if use_gn else None
    img, nc, c = ops.blocked_image(x1, x2, gn=gn, act=act, sums=True)
    x = torch.cat([x1, x2], 1).cpu() if C2 else x1.cpu()
    ref = x
    if use_gn:
        ref = ref * gn[0].cpu()[:, :, None, None] + gn[1].cpu()[:, :, None, None]
    if act:
        ref = F.silu(ref)
    got = _unblock(img.cpu())
    assert float(got[:, :, 0].abs().max()) == 0 and float(got[:, :, -1].abs().max()) == 0
    assert float(got[:, :, :, 0].abs().max()) == 0 and float(got[:, :, :, -1].abs().max()) == 0
    inner = got[:, :, 1:-1, 1:-1]
    # bf16 rounding of an fp32 value that may differ from torch's in the last bit: at most one bf16 ulp apart, mostly identical
    want = _bf(ref)
    assert _relerr(inner, want) < 4e-3
    assert float((inner == want).double().mean()) > 0.98
    assert torch.allclose(nc.cpu().double(), x.double().sum((2, 3)), rtol=1e-5, atol=1e-4)
    assert torch.allclose(c.cpu().double(), x.double().sum((0, 2, 3)), rtol=1e-5, atol=1e-4)


FWD_CASES = [
    # (N, C1, C2, H, W, Cout, gn, act, temb, res)
    (1, 32, 0, 8, 32, 128, 0, 0, 0, 0),      # one tile, two chunks (the shortest K loop: no steady-state chunk)
    (2, 64, 0, 16, 32, 128, 1, 1, 1, 1),     # four chunks, two row tiles, every epilogue term
    (1, 48, 16, 8, 64, 256, 1, 1, 0, 1),     # virtual concat resolved by the image writer, two cout tiles, two column tiles
    (1, 96, 0, 24, 32, 128, 1, 0, 1, 0),     # six chunks: steady state + both tail modes, three row tiles
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", FWD_CASES, ids=[str(i) for i in range(len(FWD_CASES))])
def test_conv_bf16_blocked_forward(backend, case):
    dev = select(backend)
    from audiodiffusion import ops
    Nn, C1, C2, H, W, Cout, use_gn, act, use_temb, use_res = case
    x1 = _rand((Nn, C1, H, W), 1, dev)
    x2 = _rand((Nn, C2, H, W), 2, dev) if C2 else None
    Ct = C1 + C2
    w = _rand((Cout, Ct, 3, 3), 3, dev, scale=(Ct * 9) ** -0.5)
    b = _rand((Cout,), 4, dev)
    gamma, beta = _rand((Ct,), 5, dev), _rand((Ct,), 6, dev)
    gn = ops.groupnorm_stats(x1, gamma, beta, 32 if Ct % 32 == 0 else 16, 1e-5, x2=x2) if use_gn else None
    temb = _rand((Nn, Cout), 7, dev) if use_temb else None
    res = _rand((Nn, Cout, H, W), 8, dev) if use_res else None
    img = ops.blocked_image(x1, x2, gn=gn, act=bool(act))
    out, st = ops.conv2d_bf16_blocked(img, ops.pack_bf16_weight(w), Cout, bias=b, chan_add=temb, residual=res, stats=True)
    # GroupNorm partial sums of the output from the epilogue: per 8x32-pixel tile (sum, sum of squares) of the values it stored
    tiles = out.cpu().double().reshape(Nn, Cout, H // 8, 8, W // 32, 32).permute(0, 1, 2, 4, 3, 5).reshape(Nn, Cout, -1, 256)
    assert torch.allclose(st.cpu()[..., 0], tiles.sum(-1), rtol=1e-5, atol=1e-4)
    assert torch.allclose(st.cpu()[..., 1], (tiles * tiles).sum(-1), rtol=1e-5, atol=1e-4)
    c = lambda t: None if t is None else t.cpu()  # noqa: E731
    tail = c(b).double()[None, :, None, None]
    if temb is not None:
        tail = tail + c(temb).double()[:, :, None, None]
    if res is not None:
        tail = tail + c(res).double()
    # tight: the kernel against a float64 convolution of ITS OWN operand image (halo cut off: the convolution pads)
    xa = _unblock(img.cpu())[:, :, 1:-1, 1:-1]
    exact = F.conv2d(xa, _bf(c(w)), None, padding=1) + tail
    assert _relerr(out.double(), exact) < 2e-6, _relerr(out.double(), exact)
    # loose: against the fp32 layer
    x = torch.cat([c(x1), c(x2)], 1) if C2 else c(x1)
    if use_gn:
        x = F.group_norm(x, 32 if Ct % 32 == 0 else 16, c(gamma), c(beta), 1e-5)
    if act:
        x = F.silu(x)
    full = F.conv2d(x.double(), c(w).double(), None, padding=1) + tail
    assert _relerr(out.double(), full) < 8e-3, _relerr(out.double(), full)


@pytest.mark.parametrize("backend", BACKENDS)
def test_conv_bf16_blocked_data_gradient(backend):
    """Backward-data pass of a 3x3 stride-1 Conv2d: the same kernel on the image of dy with transposed, flipped filters; the
    residual port accumulates into a gradient that already holds a contribution (out aliases residual)."""
    dev = select(backend)
    from audiodiffusion import ops
    Nn, Cin, Cout, H, W = 2, 128, 32, 8, 32
    w = _rand((Cout, Cin, 3, 3), 11, dev, scale=(Cin * 9) ** -0.5)
    dy = _rand((Nn, Cout, H, W), 12, dev)
    acc = _rand((Nn, Cin, H, W), 13, dev)
    img = ops.blocked_image(dy)
    dx = ops.conv2d_bf16_blocked(img, ops.pack_bf16_weight(w, transposed=True), Cin, residual=acc)
    exact = torch.nn.grad.conv2d_input((Nn, Cin, H, W), _bf(w.cpu()), _bf(dy.cpu()), padding=1) + acc.cpu().double()
    assert _relerr(dx.double(), exact) < 2e-6, _relerr(dx.double(), exact)


WGRAD_CASES = [
    # (N, Ct, H, W, Cout, max_tiles_note)
    (1, 64, 4, 32, 128),       # one tile
    (2, 64, 8, 64, 128),       # 8 tiles
    (1, 128, 16, 32, 256),     # two cin blocks x two cout tiles
    (3, 64, 4, 32, 128),       # a workgroup walks several images when the split is capped
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", WGRAD_CASES, ids=[str(i) for i in range(len(WGRAD_CASES))])
def test_conv_wgrad_bf16_blocked(backend, case):
    dev = select(backend)
    from audiodiffusion import ops
    Nn, Ct, H, W, Cout = case
    x = _rand((Nn, Ct, H, W), 1, dev)
    dy = _rand((Nn, Cout, H, W), 2, dev)
    gamma, beta = _rand((Ct,), 5, dev), _rand((Ct,), 6, dev)
    gn = ops.groupnorm_stats(x, gamma, beta, 32, 1e-5)
    x_img = ops.blocked_image(x, gn=gn, act=True)
    dy_img = ops.blocked_image(dy)
    dW = ops.conv2d_wgrad_bf16_blocked(x_img, dy_img)
    xa = _unblock(x_img.cpu())[:, :, 1:-1, 1:-1]
    exact = torch.nn.grad.conv2d_weight(xa, (Cout, Ct, 3, 3), _bf(dy.cpu()), padding=1)
    assert _relerr(dW.double(), exact) < 2e-6, _relerr(dW.double(), exact)
    full = torch.nn.grad.conv2d_weight(F.silu(F.group_norm(x.cpu(), 32, gamma.cpu(), beta.cpu(), 1e-5)).double(), (Cout, Ct, 3, 3),
                                       dy.cpu().double(), padding=1)
    assert _relerr(dW.double(), full) < 8e-3, _relerr(dW.double(), full)


@pytest.mark.parametrize("backend", BACKENDS)
def test_conv_bf16_blocked_upsample_folded_forward_and_weight_gradient(backend):
    """Upsample2D.conv (scripts/train_unet.py UpBlock2D: nearest x2, then 3x3): the blocked image is the half-resolution tensor, the
    x2 lives in the patch addresses of both kernels; the halo of the small image is the zero padding of the upsampled one."""
    dev = select(backend)
    from audiodiffusion import ops
    Nn, C, Cout, H, W = 2, 64, 128, 8, 16                     # output 16 x 32
    x = _rand((Nn, C, H, W), 1, dev)
    w = _rand((Cout, C, 3, 3), 3, dev, scale=(C * 9) ** -0.5)
    b = _rand((Cout,), 4, dev)
    dy = _rand((Nn, Cout, 2 * H, 2 * W), 5, dev)
    img = ops.blocked_image(x)
    out = ops.conv2d_bf16_blocked(img, ops.pack_bf16_weight(w), Cout, bias=b, up=True)
    xu = F.interpolate(_bf(x.cpu()), scale_factor=2.0, mode="nearest")
    exact = F.conv2d(xu, _bf(w.cpu()), b.cpu().double(), padding=1)
    assert out.shape == exact.shape
    assert _relerr(out.double(), exact) < 2e-6, _relerr(out.double(), exact)
    dW = ops.conv2d_wgrad_bf16_blocked(img, ops.blocked_image(dy), up=True)
    exact_w = torch.nn.grad.conv2d_weight(xu, (Cout, C, 3, 3), _bf(dy.cpu()), padding=1)
    assert _relerr(dW.double(), exact_w) < 2e-6, _relerr(dW.double(), exact_w)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("pad", [1, 0], ids=["padding-1-unet", "pad-0101-vae"])
def test_conv_bf16_blocked_stride_2_all_three_passes(backend, pad):
    """Downsample2D.conv — 3x3 stride 2 with padding 1 (UNet2DModel's DownBlock2D) or after pad (0, 1, 0, 1) without padding (the
    AutoencoderKL encoder): forward = every other pixel (even / odd) of the stride-1 convolution of the same blocked image; backward =
    the plain stride-1 kernels on the zero-inserted image of dy (dy(y, x) on pixel (2y, 2x) / (2y + 1, 2x + 1))."""
    dev = select(backend)
    from audiodiffusion import ops
    Nn, C, Cout, H, W = 2, 128, 128, 16, 64                    # output 8 x 32
    x = _rand((Nn, C, H, W), 1, dev)
    w = _rand((Cout, C, 3, 3), 3, dev, scale=(C * 9) ** -0.5)
    b = _rand((Cout,), 4, dev)
    dy = _rand((Nn, Cout, H // 2, W // 2), 5, dev)
    img = ops.blocked_image(x)
    out = ops.conv2d_bf16_blocked(img, ops.pack_bf16_weight(w), Cout, bias=b, up=3 if pad else 2)
    # both as an unpadded stride-2 convolution of an explicitly padded input: (1, 0, 1, 0)-padded on the top / left for padding 1 (its
    # bottom / right padding row is never read at even sizes), (0, 1, 0, 1) for the other
    xp = F.pad(_bf(x.cpu()), (1, 0, 1, 0) if pad else (0, 1, 0, 1))
    exact = F.conv2d(xp, _bf(w.cpu()), b.cpu().double(), stride=2)
    assert torch.allclose(exact, F.conv2d(_bf(x.cpu()), _bf(w.cpu()), b.cpu().double(), stride=2, padding=1)) or not pad
    assert out.shape == exact.shape
    assert _relerr(out.double(), exact) < 2e-6, _relerr(out.double(), exact)
    dimg, nc, c = ops.blocked_image(dy, zero_insert=2 if pad else 1, sums=True)
    assert tuple(dimg.shape) == (Nn, Cout // 8, H + 2, W + 2, 8)
    z = _unblock(dimg.cpu())[:, :, 1:-1, 1:-1]
    o = 0 if pad else 1
    assert torch.equal(z[:, :, o::2, o::2], _bf(dy.cpu()))
    assert float(z[:, :, 1 - o::2, :].abs().max()) == 0 and float(z[:, :, :, 1 - o::2].abs().max()) == 0
    assert torch.allclose(c.cpu().double(), dy.cpu().double().sum((0, 2, 3)), rtol=1e-5, atol=1e-4)
    # data gradient: autograd of the explicitly padded stride-2 convolution, cropped back to the input
    dx = ops.conv2d_bf16_blocked(dimg, ops.pack_bf16_weight(w, transposed=True), C)
    full = torch.nn.grad.conv2d_input((Nn, C, H + 1, W + 1), _bf(w.cpu()), _bf(dy.cpu()), stride=2)
    want = full[:, :, 1:, 1:] if pad else full[:, :, :H, :W]
    assert _relerr(dx.double(), want) < 2e-6, _relerr(dx.double(), want)
    dW = ops.conv2d_wgrad_bf16_blocked(img, dimg)
    want_w = torch.nn.grad.conv2d_weight(xp, (Cout, C, 3, 3), _bf(dy.cpu()), stride=2)
    assert _relerr(dW.double(), want_w) < 2e-6, _relerr(dW.double(), want_w)


NARROW_CASES = [
    # (N, Cin, H, W, Cout, temb, res, up)        rows of 16 / 8 pixels: 2 / 4 images side by side in one 32-column tile, K split
    (2, 64, 16, 16, 128, 1, 1, 0),      # one pair, two row tiles, 4 chunks -> 2 parts
    (4, 128, 8, 16, 128, 0, 0, 0),      # two pairs, 8 chunks -> 4 parts
    (4, 64, 8, 8, 128, 1, 1, 0),        # one group of four 8x8 images, 4 chunks -> 2 parts
    (8, 256, 8, 8, 256, 0, 1, 0),       # two groups, two cout tiles, 16 chunks -> 8 parts
    (2, 64, 16, 16, 128, 0, 0, 1),      # Upsample2D.conv 8x8 -> 16x16: nearest x2 folded into the patch addresses of each image
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", NARROW_CASES, ids=[str(i) for i in range(len(NARROW_CASES))])
def test_conv_bf16_blocked_narrow_rows_tile_images_side_by_side(backend, case):
    """The 16x16 and 8x8 levels of the 256x256 model (round 4): the same kernel, a tile's 32 columns = 2 / 4 images that share their
    zero halo columns; K split over a layer-determined number of workgroups, slabs added in order by the split-K finish pass with
    bias / per-sample term / residual. Row-independence: a sample's result does not depend on which images share its tile."""
    dev = select(backend)
    from audiodiffusion import ops
    Nn, Cin, H, W, Cout, use_temb, use_res, up = case
    Hs, Ws = (H // 2, W // 2) if up else (H, W)
    x = _rand((Nn, Cin, Hs, Ws), 1, dev)
    w = _rand((Cout, Cin, 3, 3), 3, dev, scale=(Cin * 9) ** -0.5)
    b = _rand((Cout,), 4, dev)
    temb = _rand((Nn, Cout), 7, dev) if use_temb else None
    res = _rand((Nn, Cout, H, W), 8, dev) if use_res else None
    img = ops.blocked_image(x)
    wb = ops.pack_bf16_weight(w)
    out = ops.conv2d_bf16_blocked(img, wb, Cout, bias=b, chan_add=temb, residual=res, up=bool(up))
    c = lambda t: None if t is None else t.cpu()  # noqa: E731
    tail = c(b).double()[None, :, None, None]
    if temb is not None:
        tail = tail + c(temb).double()[:, :, None, None]
    if res is not None:
        tail = tail + c(res).double()
    xa = _unblock(img.cpu())[:, :, 1:-1, 1:-1]
    if up:
        xa = F.interpolate(xa, scale_factor=2, mode="nearest")
    exact = F.conv2d(xa, _bf(c(w)), None, padding=1) + tail
    assert out.shape == exact.shape
    assert _relerr(out.double(), exact) < 2e-6, _relerr(out.double(), exact)
    # the same samples in another order land in other tile slots: bit-identical rows
    perm = torch.arange(Nn - 1, -1, -1)
    out_p = ops.conv2d_bf16_blocked(ops.blocked_image(x[perm.to(x.device)]), wb, Cout, bias=b,
                                    chan_add=None if temb is None else temb[perm.to(x.device)],
                                    residual=None if res is None else res[perm.to(x.device)], up=bool(up))
    assert torch.equal(out_p.cpu()[perm], out.cpu())


NARROW_WGRAD_CASES = [
    # (N, Ct, H, W, Cout, up)
    (2, 64, 16, 16, 128, 0),      # one pair of 16x16 images: 4 tiles
    (4, 128, 8, 16, 128, 0),      # two pairs, two cin blocks
    (4, 64, 8, 8, 128, 0),        # one group of four 8x8 images: 2 tiles (the k-step's two 8-pixel halves are two images)
    (8, 64, 8, 8, 256, 0),        # two groups, two cout tiles
    (2, 64, 16, 16, 128, 1),      # Upsample2D.conv 8x8 -> 16x16
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", NARROW_WGRAD_CASES, ids=[str(i) for i in range(len(NARROW_WGRAD_CASES))])
def test_conv_wgrad_bf16_blocked_narrow_rows(backend, case):
    dev = select(backend)
    from audiodiffusion import ops
    Nn, Ct, H, W, Cout, up = case
    Hs, Ws = (H // 2, W // 2) if up else (H, W)
    x = _rand((Nn, Ct, Hs, Ws), 1, dev)
    dy = _rand((Nn, Cout, H, W), 2, dev)
    x_img = ops.blocked_image(x)
    dy_img = ops.blocked_image(dy)
    dW = ops.conv2d_wgrad_bf16_blocked(x_img, dy_img, up=bool(up))
    xa = _unblock(x_img.cpu())[:, :, 1:-1, 1:-1]
    if up:
        xa = F.interpolate(xa, scale_factor=2, mode="nearest")
    dyb = _unblock(dy_img.cpu())[:, :, 1:-1, 1:-1]
    exact = torch.nn.grad.conv2d_weight(xa, (Cout, Ct, 3, 3), dyb, padding=1)
    assert _relerr(dW.double(), exact) < 2e-6, _relerr(dW.double(), exact)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("shape", [(1, 64, 8, 32), (2, 64, 16, 16), (4, 64, 8, 8)], ids=["rows32", "rows16", "rows8"])
def test_blocked_kernels_on_binary16_operands(backend, shape):
    """`--mixed_precision fp16` (train_unet.py:391-395): option conv_op16_f16 = 1 makes the operand format of the image writer, the
    forward / data-gradient kernel and the weight-gradient kernel IEEE binary16 (template flag; same staging, `v_mfma_f32_32x32x16_f16`).
    Tight bars against float64 convolutions of the kernels' OWN operand images read as binary16, on the 32-pixel-row tiling and on
    both narrow-row tilings."""
    dev = select(backend)
    from audiodiffusion import _native, ops
    Nn, C, H, W = shape
    Cout = 128
    x = _rand((Nn, C, H, W), 1, dev)
    dy = _rand((Nn, Cout, H, W), 2, dev)
    w = _rand((Cout, C, 3, 3), 3, dev, scale=(C * 9) ** -0.5)
    lib = _native.lib()

    def as_f16(img):            # the blocked image tensor is typed bfloat16; under the option its bits are binary16
        n, cg, hp, wp, _ = img.shape
        return img.cpu().view(torch.float16).permute(0, 1, 4, 2, 3).reshape(n, cg * 8, hp, wp).to(torch.float64)[:, :, 1:-1, 1:-1]

    _native.check(lib.adm_set_option(b"conv_op16_f16", 1))
    try:
        x_img, dy_img = ops.blocked_image(x), ops.blocked_image(dy)
        wb = ops.pack_bf16_weight(w)
        out = ops.conv2d_bf16_blocked(x_img, wb, Cout)
        dW = ops.conv2d_wgrad_bf16_blocked(x_img, dy_img)
    finally:
        _native.check(lib.adm_set_option(b"conv_op16_f16", 0))
    xa, dya = as_f16(x_img), as_f16(dy_img)
    assert float((xa - x.cpu().half().double()).abs().max()) == 0.0          # round-to-nearest-even binary16 of the input
    w16 = w.cpu().half().double()
    assert _relerr(out.double(), F.conv2d(xa, w16, None, padding=1)) < 2e-6
    assert _relerr(dW.double(), torch.nn.grad.conv2d_weight(xa, (Cout, C, 3, 3), dya, padding=1)) < 2e-6
