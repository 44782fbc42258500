"""Mixed-precision 3x3 convolution (`--mixed_precision bf16`, scripts/train_unet.py:391-401): bf16 MFMA operands, fp32
accumulation and epilogue.  Two bars per case: (tight) against a float64 convolution of the SAME bf16-rounded operands —
what the kernel is meant to compute, only the accumulation order differs; (loose) against the plain fp32 convolution —
the error the mixed-precision mode introduces, bounded by bf16's 2^-9 relative rounding of each operand."""
import pytest
import torch
import torch.nn.functional as F

from native_backend import BACKENDS, select
from test_kernels import _rand, _relerr

CASES = [
    # (N, C1, C2, H, W, Cout, up, gn, act, temb, res)
    (1, 32, 0, 16, 16, 128, 0, 0, 0, 0, 0),     # bare convolution: operands exactly representable after rounding
    (2, 32, 0, 16, 32, 128, 0, 1, 1, 1, 1),     # GroupNorm + SiLU load path, all epilogue terms, two column tiles
    (1, 48, 16, 32, 16, 256, 0, 1, 1, 0, 1),    # virtual concat (seam on an odd chunk boundary), two cout tiles, two row tiles
    (1, 32, 0, 8, 8, 128, 1, 1, 1, 1, 0),       # nearest x2 folded into the load path
    (3, 96, 0, 16, 16, 128, 0, 1, 0, 0, 0),     # six chunks, GroupNorm without activation
]


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float64)


def _refs(x1, x2, w, b, up, gn, act, temb, res):
    x = torch.cat([x1, x2], 1) if x2 is not None else x1
    if gn is not None:
        x = F.group_norm(x, 32, gn[0], gn[1], 1e-5)
    if act:
        x = F.silu(x)
    if up:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    tail = b.double()[None, :, None, None]
    if temb is not None:
        tail = tail + temb.double()[:, :, None, None]
    if res is not None:
        tail = tail + res.double()
    exact = F.conv2d(_bf(x), _bf(w), None, padding=1) + tail
    full = F.conv2d(x.double(), w.double(), None, padding=1) + tail
    return exact, full


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", CASES, ids=[str(i) for i in range(len(CASES))])
def test_conv_bf16_forward(backend, case):
    dev = select(backend)
    from audiodiffusion import _native, ops
    Nn, C1, C2, H, W, Cout, up, use_gn, act, use_temb, use_res = case
    x1 = _rand((Nn, C1, H, W), 1, dev)
    x2 = _rand((Nn, C2, H, W), 2, dev) if C2 else None
    Ct = C1 + C2
    w = _rand((Cout, Ct, 3, 3), 3, dev, scale=(Ct * 9) ** -0.5)
    b = _rand((Cout,), 4, dev)
    gamma, beta = _rand((Ct,), 5, dev), _rand((Ct,), 6, dev)
    gn = ops.groupnorm_stats(x1, gamma, beta, 32 if Ct % 32 == 0 else 16, 1e-5, x2=x2) if use_gn else None
    temb = _rand((Nn, Cout), 7, dev) if use_temb else None
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    res = _rand((Nn, Cout, Ho, Wo), 8, dev) if use_res else None
    _native.check(_native.lib().adm_set_option(b"conv_bf16", 1))
    try:
        out = ops.conv2d(x1, ops.pack_conv_weight(w), b, 3, x2=x2, up=bool(up), gn=gn, act=bool(act), chan_add=temb,
                         residual=res, wino=ops.pack_winograd_weight(w), bf16=ops.pack_bf16_weight(w))
        assert _native.lib().adm_last_conv_variant() == 5316, "the bf16 kernel was not selected"
    finally:
        _native.check(_native.lib().adm_set_option(b"conv_bf16", 0))
    c = lambda t: None if t is None else t.cpu()  # noqa: E731
    if use_gn:   # the reference normalises with the same group count
        groups = 32 if Ct % 32 == 0 else 16
        xg = torch.cat([c(x1), c(x2)], 1) if C2 else c(x1)
        xn = F.group_norm(xg, groups, c(gamma), c(beta), 1e-5)
        exact, full = _refs(xn, None, c(w), c(b), up, None, act, c(temb), c(res))
    else:
        exact, full = _refs(c(x1), c(x2), c(w), c(b), up, None, act, c(temb), c(res))
    assert out.shape == exact.shape
    # tight: identical operands. Without GroupNorm/SiLU the rounding points coincide exactly; with them an fp32 last-bit
    # difference of the activation can flip a bf16 rounding (2^-9 of one operand among Cin*9 products)
    tight = 2e-6 if not (use_gn or act) else 3e-4
    assert _relerr(out.double(), exact) < tight, _relerr(out.double(), exact)
    # loose: the mixed-precision error itself
    assert _relerr(out.double(), full) < 8e-3, _relerr(out.double(), full)


@pytest.mark.parametrize("backend", BACKENDS)
def test_conv_bf16_data_gradient(backend):
    """Backward-data pass of a 3x3 stride-1 Conv2d as the same kernel on dy with transposed, flipped bf16 filters."""
    dev = select(backend)
    from audiodiffusion import _native, ops
    Nn, Cin, Cout, H, W = 2, 128, 32, 16, 32
    w = _rand((Cout, Cin, 3, 3), 11, dev, scale=(Cin * 9) ** -0.5)
    dy = _rand((Nn, Cout, H, W), 12, dev)
    acc = _rand((Nn, Cin, H, W), 13, dev)          # gradient already accumulated in dx (residual fan-in)
    _native.check(_native.lib().adm_set_option(b"conv_bf16", 1))
    try:
        dx = ops.conv2d(dy, ops.pack_conv_weight_T(w), None, 3, residual=acc, wino=ops.pack_winograd_weight_T(w),
                        bf16=ops.pack_bf16_weight(w, transposed=True))
        assert _native.lib().adm_last_conv_variant() == 5316
    finally:
        _native.check(_native.lib().adm_set_option(b"conv_bf16", 0))
    exact = torch.nn.grad.conv2d_input((Nn, Cin, H, W), _bf(w.cpu()), _bf(dy.cpu()), padding=1) + acc.cpu().double()
    assert _relerr(dx.double(), exact) < 2e-6, _relerr(dx.double(), exact)


@pytest.mark.parametrize("backend", BACKENDS)
def test_bf16_option_off_keeps_fp32(backend):
    dev = select(backend)
    from audiodiffusion import _native, ops
    x = _rand((1, 32, 16, 16), 1, dev)
    w = _rand((128, 32, 3, 3), 2, dev, scale=0.1)
    ops.conv2d(x, ops.pack_conv_weight(w), None, 3, wino=ops.pack_winograd_weight(w), bf16=ops.pack_bf16_weight(w))
    assert _native.lib().adm_last_conv_variant() != 5316


WGRAD_CASES = [
    # (N, C1, C2, H, W, Cout, up, gn, act, max_split)
    (1, 32, 0, 4, 16, 128, 0, 0, 0, 0),       # one tile, bare: rounding points coincide exactly
    (2, 32, 0, 8, 32, 128, 0, 1, 1, 0),       # GroupNorm + SiLU recomputed on the load path, 8 tiles
    (1, 32, 32, 16, 16, 256, 0, 1, 1, 2),     # virtual concat, two cout tiles, several tiles per workgroup (split-K 2)
    (1, 64, 0, 8, 8, 128, 1, 1, 1, 0),        # nearest x2 folded
    (3, 32, 0, 4, 16, 128, 0, 0, 0, 1),       # one workgroup walks three images
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", WGRAD_CASES, ids=[str(i) for i in range(len(WGRAD_CASES))])
def test_conv_bf16_weight_gradient(backend, case):
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
    exact = torch.nn.grad.conv2d_weight(_bf(a), (Cout, Ct, 3, 3), _bf(dy), padding=1)
    full = torch.nn.grad.conv2d_weight(a.double(), (Cout, Ct, 3, 3), dy.double(), padding=1)
    gn = ops.groupnorm_stats(x1, gamma, beta, 32, 1e-5, x2=x2) if use_gn else None
    _native.check(_native.lib().adm_set_option(b"conv_bf16", 1))
    _native.check(_native.lib().adm_set_option(b"wgrad_max_split", max_split))
    try:
        dW = ops.conv2d_wgrad(x1, dy.to(dev), Cout, 3, x2=x2, up=bool(up), gn=gn, act=bool(act))
    finally:
        _native.check(_native.lib().adm_set_option(b"conv_bf16", 0))
        _native.check(_native.lib().adm_set_option(b"wgrad_max_split", 0))
    tight = 2e-6 if not (use_gn or act) else 3e-4
    assert _relerr(dW.double(), exact) < tight, _relerr(dW.double(), exact)
    assert _relerr(dW.double(), full) < 8e-3, _relerr(dW.double(), full)
    # and it really was the bf16 kernel: the fp32 path would match `full` to 1e-5 but not `exact`
    assert _relerr(dW.double(), full) > 1e-5


# ---------------------------------------------------------------- 1x1 convolutions on bf16 operands (option conv_bf16 = 2)
PW_CASES = [
    # (N, C1, C2, H, W, Cout, gn, act, temb, res)
    (1, 32, 0, 16, 16, 128, 0, 0, 0, 0),      # bare shortcut convolution
    (2, 64, 32, 16, 32, 128, 0, 0, 0, 1),     # shortcut on a virtual concat (seam on a chunk boundary) + residual, 2 tiles
    (2, 64, 0, 16, 16, 256, 1, 0, 0, 0),      # attention projection: GroupNorm (no SiLU) on the load path, two cout tiles
    (1, 32, 0, 16, 16, 128, 1, 1, 1, 1),      # all load-path and epilogue terms
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", PW_CASES, ids=[str(i) for i in range(len(PW_CASES))])
def test_conv1x1_bf16_forward(backend, case):
    dev = select(backend)
    from audiodiffusion import _native, ops
    Nn, C1, C2, H, W, Cout, use_gn, act, use_temb, use_res = case
    x1 = _rand((Nn, C1, H, W), 1, dev)
    x2 = _rand((Nn, C2, H, W), 2, dev) if C2 else None
    Ct = C1 + C2
    w = _rand((Cout, Ct, 1, 1), 3, dev, scale=Ct ** -0.5)
    b = _rand((Cout,), 4, dev)
    gamma, beta = _rand((Ct,), 5, dev), _rand((Ct,), 6, dev)
    gn = ops.groupnorm_stats(x1, gamma, beta, 32, 1e-5, x2=x2) if use_gn else None
    temb = _rand((Nn, Cout), 7, dev) if use_temb else None
    res = _rand((Nn, Cout, H, W), 8, dev) if use_res else None
    _native.check(_native.lib().adm_set_option(b"conv_bf16", 2))
    try:
        out = ops.conv2d(x1, ops.pack_conv_weight(w), b, 1, x2=x2, pad_lo=0, gn=gn, act=bool(act), chan_add=temb,
                         residual=res, bf16=ops.pack_bf16_weight(w))
        assert _native.lib().adm_last_conv_variant() == 5116, "the 1x1 bf16 kernel was not selected"
        _native.check(_native.lib().adm_set_option(b"conv_bf16", 1))     # mode 1 keeps 1x1 convolutions in fp32
        ops.conv2d(x1, ops.pack_conv_weight(w), b, 1, x2=x2, pad_lo=0, gn=gn, act=bool(act), chan_add=temb,
                   residual=res, bf16=ops.pack_bf16_weight(w))
        assert _native.lib().adm_last_conv_variant() != 5116
    finally:
        _native.check(_native.lib().adm_set_option(b"conv_bf16", 0))
    c = lambda t: None if t is None else t.cpu()  # noqa: E731
    x = torch.cat([c(x1), c(x2)], 1) if C2 else c(x1)
    if use_gn:
        x = F.group_norm(x, 32, c(gamma), c(beta), 1e-5)
    if act:
        x = F.silu(x)
    tail = c(b).double()[None, :, None, None]
    if use_temb:
        tail = tail + c(temb).double()[:, :, None, None]
    if use_res:
        tail = tail + c(res).double()
    exact = F.conv2d(_bf(x), _bf(c(w))) + tail
    full = F.conv2d(x.double(), c(w).double()) + tail
    tight = 2e-6 if not (use_gn or act) else 3e-4
    assert _relerr(out.double(), exact) < tight, _relerr(out.double(), exact)
    assert _relerr(out.double(), full) < 8e-3


@pytest.mark.parametrize("backend", BACKENDS)
def test_conv1x1_bf16_data_gradient(backend):
    dev = select(backend)
    from audiodiffusion import _native, ops
    Nn, Cin, Cout, H, W = 2, 128, 64, 16, 16
    w = _rand((Cout, Cin, 1, 1), 11, dev, scale=Cin ** -0.5)
    dy = _rand((Nn, Cout, H, W), 12, dev)
    _native.check(_native.lib().adm_set_option(b"conv_bf16", 2))
    try:
        dx = ops.conv2d(dy, ops.pack_conv_weight_T(w), None, 1, pad_lo=0, bf16=ops.pack_bf16_weight(w, transposed=True))
        assert _native.lib().adm_last_conv_variant() == 5116
    finally:
        _native.check(_native.lib().adm_set_option(b"conv_bf16", 0))
    exact = torch.nn.grad.conv2d_input((Nn, Cin, H, W), _bf(w.cpu()), _bf(dy.cpu()))
    assert _relerr(dx.double(), exact) < 2e-6


PW_WGRAD_CASES = [
    # (N, C1, C2, H, W, Cout, gn, act, max_split)
    (1, 128, 0, 8, 8, 128, 0, 0, 0),       # one stage, bare
    (2, 128, 128, 8, 16, 128, 0, 0, 2),    # shortcut on a virtual concat, several stages per workgroup
    (3, 128, 0, 8, 8, 256, 1, 1, 1),       # GroupNorm + SiLU recomputed on the load path, odd stage count, two cout tiles
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", PW_WGRAD_CASES, ids=[str(i) for i in range(len(PW_WGRAD_CASES))])
def test_conv1x1_bf16_weight_gradient(backend, case):
    dev = select(backend)
    from audiodiffusion import _native, ops
    Nn, C1, C2, H, W, Cout, use_gn, act, max_split = case
    x1 = _rand((Nn, C1, H, W), 1, dev)
    x2 = _rand((Nn, C2, H, W), 2, dev) if C2 else None
    Ct = C1 + C2
    gamma, beta = _rand((Ct,), 5, dev), _rand((Ct,), 6, dev)
    a = torch.cat([x1, x2], 1).cpu() if C2 else x1.cpu()
    if use_gn:
        a = F.group_norm(a, 32, gamma.cpu(), beta.cpu(), 1e-5)
    if act:
        a = F.silu(a)
    dy = _rand((Nn, Cout, H, W), 7, "cpu")
    exact = torch.nn.grad.conv2d_weight(_bf(a), (Cout, Ct, 1, 1), _bf(dy))
    full = torch.nn.grad.conv2d_weight(a.double(), (Cout, Ct, 1, 1), dy.double())
    gn = ops.groupnorm_stats(x1, gamma, beta, 32, 1e-5, x2=x2) if use_gn else None
    _native.check(_native.lib().adm_set_option(b"conv_bf16", 2))
    _native.check(_native.lib().adm_set_option(b"wgrad_max_split", max_split))
    try:
        dW = ops.conv2d_wgrad(x1, dy.to(dev), Cout, 1, x2=x2, pad_lo=0, gn=gn, act=bool(act))
    finally:
        _native.check(_native.lib().adm_set_option(b"conv_bf16", 0))
        _native.check(_native.lib().adm_set_option(b"wgrad_max_split", 0))
    tight = 2e-6 if not (use_gn or act) else 3e-4
    assert _relerr(dW.double(), exact) < tight, _relerr(dW.double(), exact)
    assert 1e-5 < _relerr(dW.double(), full) < 8e-3


# ---------------------------------------------------------------- tile-boundary cases (several pixel / cout tiles, ragged grids)
PERSIST_CASES = [
    # (N, C1, C2, H, W, Cout, up, gn, act, temb, res)
    (2, 64, 0, 16, 32, 128, 0, 1, 1, 1, 1),      # 4 tiles on a 3-workgroup grid: tile boundaries inside a workgroup, 4 chunks
    (3, 48, 16, 32, 16, 256, 0, 1, 1, 0, 1),     # two cout tiles per pixel tile (filters change across the boundary), 3 images
    (2, 128, 0, 16, 16, 128, 0, 0, 0, 0, 0),     # one tile per image: the GroupNorm row set flips at every boundary, 8 chunks
    (1, 64, 0, 16, 16, 128, 1, 1, 1, 0, 0),      # nearest x2 folded: 32x32 output, image borders on every tile
    (5, 64, 0, 16, 16, 128, 0, 1, 0, 1, 0),      # 5 tiles on 3 workgroups: ranges of 2, 2 and 1 tiles
]


@pytest.mark.parametrize("backend", BACKENDS)
def test_conv_bf16_stride2_data_gradient_by_zero_insertion(backend):
    """Backward-data pass of a 3x3 stride-2 Conv2d (Downsample2D) = the bf16 kernel on the zero-inserted dy (up = 2) with
    transposed, flipped filters; mixed-precision level 2."""
    dev = select(backend)
    from audiodiffusion import _native, ops
    Nn, Cin, Cout, H, W = 2, 128, 64, 32, 32           # forward: (Cin, 32, 32) -> (Cout, 16, 16)
    w = _rand((Cout, Cin, 3, 3), 11, dev, scale=(Cin * 9) ** -0.5)
    dy = _rand((Nn, Cout, H // 2, W // 2), 12, dev)
    _native.check(_native.lib().adm_set_option(b"conv_bf16", 2))
    try:
        dx = ops.conv2d(dy, ops.pack_conv_weight_T(w), None, 3, up=2, bf16=ops.pack_bf16_weight(w, transposed=True))
        assert _native.lib().adm_last_conv_variant() == 5316
        _native.check(_native.lib().adm_set_option(b"conv_bf16", 1))        # level 1 keeps this one in fp32
        ops.conv2d(dy, ops.pack_conv_weight_T(w), None, 3, up=2, bf16=ops.pack_bf16_weight(w, transposed=True))
        assert _native.lib().adm_last_conv_variant() != 5316
    finally:
        _native.check(_native.lib().adm_set_option(b"conv_bf16", 0))
    exact = torch.nn.grad.conv2d_input((Nn, Cin, H, W), _bf(w.cpu()), _bf(dy.cpu()), stride=2, padding=1)
    assert _relerr(dx.double(), exact) < 2e-6, _relerr(dx.double(), exact)


# ---------------------------------------------------------------- the same kernels on IEEE binary16 operands (`--mixed_precision fp16`)
def _h(t):
    return t.to(torch.float16).to(torch.float64)


@pytest.mark.parametrize("backend", BACKENDS)
def test_fp16_operand_format_forward_wgrad_and_1x1(backend):
    """Option conv_op16_f16 = 1 switches the 16-bit-operand kernels from bf16 to binary16 (template flag: same staging, same
    MFMA shape, `v_cvt_f16_f32` + `v_mfma_f32_32x32x16_f16`). Bare convolutions (no GroupNorm / SiLU): the result must equal the
    float64 convolution of the fp16-rounded operands to accumulation-order precision, and differ from the bf16 result."""
    dev = select(backend)
    from audiodiffusion import _native, ops
    lib = _native.lib()
    Nn, Ci, Co, H, W = 2, 32, 128, 16, 16
    x = _rand((Nn, Ci, H, W), 1, dev)
    w = _rand((Co, Ci, 3, 3), 3, dev, scale=(Ci * 9) ** -0.5)
    b = _rand((Co,), 4, dev)
    dy = _rand((Nn, Co, H, W), 7, dev)
    w1 = _rand((128, 128, 1, 1), 8, dev, scale=128 ** -0.5)
    x1 = _rand((Nn, 128, 16, 16), 9, dev)
    out = {}
    for f16 in (0, 1):
        _native.check(lib.adm_set_option(b"conv_op16_f16", f16))
        _native.check(lib.adm_set_option(b"conv_bf16", 2))
        try:
            y = ops.conv2d(x, ops.pack_conv_weight(w), b, 3, bf16=ops.pack_bf16_weight(w))
            assert lib.adm_last_conv_variant() == 5316
            dW = ops.conv2d_wgrad(x, dy, Co, 3)
            y1 = ops.conv2d(x1, ops.pack_conv_weight(w1), None, 1, pad_lo=0, bf16=ops.pack_bf16_weight(w1))
            assert lib.adm_last_conv_variant() == 5116
            out[f16] = (y.cpu().double(), dW.cpu().double(), y1.cpu().double())
        finally:
            _native.check(lib.adm_set_option(b"conv_bf16", 0))
            _native.check(lib.adm_set_option(b"conv_op16_f16", 0))
    c = lambda t: t.cpu()  # noqa: E731
    ref_y = F.conv2d(_h(c(x)), _h(c(w)), None, padding=1) + c(b).double()[None, :, None, None]
    ref_dW = torch.nn.grad.conv2d_weight(_h(c(x)), (Co, Ci, 3, 3), _h(c(dy)), padding=1)
    ref_y1 = F.conv2d(_h(c(x1)), _h(c(w1)), None)
    for got, ref in zip(out[1], (ref_y, ref_dW, ref_y1)):
        assert _relerr(got, ref) < 2e-6, _relerr(got, ref)
    for k in range(3):
        assert 1e-5 < _relerr(out[1][k], out[0][k]) < 8e-3        # bf16 and binary16 round differently, both close to fp32


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [(64, 32, 3), (128, 96, 3), (192, 64, 1), (40, 24, 3)], ids=["one-tile", "tiles", "1x1", "gather"])
def test_bf16_filter_images_bit_for_bit(backend, case):
    """adm_pack_bf16_weight: [tap][Cin/8][Cout] x 8 cins (forward) and [flipped tap][Cout/8][Cin] x 8 couts (data gradient), each
    element the round-to-nearest-even bf16 of the fp32 weight — the LDS-tiled writer (Cout % 64 == 0, Cin % 32 == 0; round 4) and the
    per-element gather (other shapes) against the layout written out in torch."""
    dev = select(backend)
    from audiodiffusion import ops
    co, ci, ks = case
    w = _rand((co, ci, ks, ks), 21, dev) if ks == 3 else _rand((co, ci), 21, dev)
    w4 = w.cpu().reshape(co, ci, ks * ks).to(torch.bfloat16)
    fwd = ops.pack_bf16_weight(w).cpu()
    want = w4.permute(2, 1, 0).reshape(ks * ks, ci // 8, 8, co).permute(0, 1, 3, 2)
    assert torch.equal(fwd.view(torch.int16), want.contiguous().view(torch.int16))
    bwd = ops.pack_bf16_weight(w, transposed=True).cpu()
    want_t = w4.flip(2).permute(2, 0, 1).reshape(ks * ks, co // 8, 8, ci).permute(0, 1, 3, 2)
    assert torch.equal(bwd.view(torch.int16), want_t.contiguous().view(torch.int16))
