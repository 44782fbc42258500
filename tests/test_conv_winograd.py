"""Winograd F(2x2,3x3) variants of the fused 3x3 convolution (mode 4 is the default 3x3 stride-1 path) vs torch fp32."""
import pytest  # noqa: E402
import torch  # noqa: E402

from native_backend import BACKENDS, select  # noqa: E402
from test_kernels import _conv_ref, _rand, _relerr  # noqa: E402

CASES = [
    # (N, C1, C2, H, W, Cout, up, gn, act, temb, res)
    (1, 32, 0, 8, 16, 32, 0, 1, 1, 1, 1),
    (2, 32, 32, 16, 32, 64, 0, 1, 1, 0, 1),
    (1, 32, 0, 8, 8, 32, 1, 0, 0, 1, 0),      # upsample folded: 8x8 -> 16x16
    (1, 64, 0, 16, 16, 96, 0, 1, 0, 0, 0),
    (1, 32, 0, 8, 8, 64, 1, 0, 0, 1, 0),      # v2-eligible: upsample folded, 64 couts
    (2, 64, 0, 8, 16, 128, 0, 1, 1, 1, 1),    # v2-eligible: two cout tiles, all epilogue terms
    (1, 32, 0, 8, 8, 64, 1, 1, 1, 1, 0),      # v3-eligible: GroupNorm + SiLU + folded upsample
    (3, 32, 0, 16, 32, 64, 0, 1, 1, 1, 0),    # v3: 12 tiles on a 3-block persistent grid (emulator), image borders
    (2, 32, 32, 24, 48, 128, 0, 1, 1, 1, 1),  # v3: virtual concat, interior + border tiles, residual
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("mode", [4], ids=["v4regfilters"])       # (modes 1-3 were the kernel generations retired in round 6)
@pytest.mark.parametrize("case", [c for c in CASES if c[5] % 64 == 0 and (c[1] + c[2]) % 32 == 0 and not (c[8] and not c[7])],
                         ids=lambda c: "-".join(str(v) for v in c))
def test_conv_winograd(backend, case, mode):
    """conv_wino4_kernel tiles 64 output and 32 input channels (four chunks in flight) and needs GroupNorm whenever SiLU is requested."""
    dev = select(backend)
    from audiodiffusion import _native, ops
    _native.check(_native.lib().adm_set_option(b"conv_wino", mode))
    _native.check(_native.lib().adm_set_option(b"wino5", 0))            # this test pins conv_wino4_kernel (v5 has its own below)
    try:
        _run_case(dev, case, 4310 + mode)
    finally:
        _native.check(_native.lib().adm_set_option(b"conv_wino", -1))   # back to the default
        _native.check(_native.lib().adm_set_option(b"wino5", -1))


def _run_case(dev, case, want_variant):
    from audiodiffusion import _native, ops
    Nn, C1, C2, H, W, Cout, up, use_gn, act, use_temb, use_res = case
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
    out = ops.conv2d(x1, ops.pack_conv_weight(w), b, 3, x2=x2, up=bool(up), gn=gn, act=bool(act), chan_add=temb,
                     residual=res, wino=ops.pack_winograd_weight(w))
    assert _native.lib().adm_last_conv_variant() == want_variant, "the Winograd kernel was not selected"
    c = lambda t: None if t is None else t.cpu()  # noqa: E731
    ref = _conv_ref(c(x1), c(x2), c(w), c(b), 3, 1, up, (c(gamma), c(beta)) if use_gn else None, act, c(temb), c(res))
    assert out.shape == ref.shape
    assert _relerr(out, ref) < 1e-4, _relerr(out, ref)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("mode,Cin,Cout,variant", [(4, 64, 32, 4314), (4, 128, 64, 4315)], ids=["v4", "v5-128-couts"])
def test_conv_winograd_data_gradient(backend, mode, Cin, Cout, variant):
    """3x3 stride-1 backward-data pass as a Winograd convolution with transposed/flipped filters vs torch autograd
    (the data-gradient convolution maps Cout -> Cin channels, so ITS output-channel count is the forward's Cin)."""
    dev = select(backend)
    from audiodiffusion import _native, ops
    Nn, H, W = 2, 16, 32
    w = _rand((Cout, Cin, 3, 3), 11, dev, scale=(Cin * 9) ** -0.5)
    dy = _rand((Nn, Cout, H, W), 12, dev)
    acc = _rand((Nn, Cin, H, W), 13, dev)          # gradient already accumulated in dx (residual fan-in)
    _native.check(_native.lib().adm_set_option(b"conv_wino", mode))
    try:
        dx = ops.conv2d(dy, ops.pack_conv_weight_T(w), None, 3, residual=acc, wino=ops.pack_winograd_weight_T(w))
        if variant == 4315 and backend == "hip":   # 8 tiles of 128 couts do not fill 256 CUs: the launcher keeps v4's 64-cout tiles (bit-identical)
            variant = 4314
        assert _native.lib().adm_last_conv_variant() == variant, "the Winograd kernel was not selected"
    finally:
        _native.check(_native.lib().adm_set_option(b"conv_wino", -1))
    ref = torch.nn.grad.conv2d_input((Nn, Cin, H, W), w.cpu(), dy.cpu(), padding=1) + acc.cpu()
    assert _relerr(dx, ref) < 1e-4, _relerr(dx, ref)


@pytest.mark.parametrize("backend", BACKENDS)
def test_groupnorm_statistics_from_the_conv_epilogue(backend):
    """The v4 kernel's epilogue writes (sum, sum of squares) per (sample, cout, 8x16 tile) of its FINAL output (bias, per-sample
    term and residual included); adm_groupnorm_finalize on them must give what the read pass (adm_groupnorm_stats) gives on the
    stored output — one producer, and a virtual concat of two producers with different tile counts."""
    dev = select(backend)
    from audiodiffusion import _native, ops
    Nn = 2
    x = _rand((Nn, 64, 16, 32), 1, dev)
    w1 = _rand((64, 64, 3, 3), 2, dev, scale=(64 * 9) ** -0.5)
    w2 = _rand((128, 64, 3, 3), 3, dev, scale=(64 * 9) ** -0.5)
    b1, b2 = _rand((64,), 4, dev), _rand((128,), 5, dev)
    temb, res = _rand((Nn, 64), 6, dev), _rand((Nn, 64, 16, 32), 7, dev)
    gin = ops.groupnorm_stats(x, _rand((64,), 8, dev), _rand((64,), 9, dev), 32, 1e-5)
    y1, s1 = ops.conv2d(x, ops.pack_conv_weight(w1), b1, 3, gn=gin, act=True, chan_add=temb, residual=res,
                        wino=ops.pack_winograd_weight(w1), stats=True)
    assert _native.lib().adm_last_conv_variant() == 4314 and s1 is not None and tuple(s1.shape) == (Nn, 64, 4, 2)
    assert not torch.isnan(s1).any()                                         # every (sample, cout, tile) slot was written
    want = torch.stack([y1.double().reshape(Nn, 64, 2, 8, 2, 16).sum((3, 5)).reshape(Nn, 64, 4),
                        (y1.double() ** 2).reshape(Nn, 64, 2, 8, 2, 16).sum((3, 5)).reshape(Nn, 64, 4)], -1)
    assert torch.allclose(s1.cpu(), want.cpu(), rtol=2e-6, atol=2e-5)       # 8 values per lane are summed in fp32, the rest in fp64
    gamma, beta = _rand((64,), 10, dev), _rand((64,), 11, dev)
    sc, sh = ops.groupnorm_finalize(s1, gamma, beta, 32, 1e-5, 16 * 32)
    rsc, rsh = ops.groupnorm_stats(y1, gamma, beta, 32, 1e-5)
    assert torch.allclose(sc.cpu(), rsc.cpu(), rtol=2e-6, atol=1e-7) and torch.allclose(sh.cpu(), rsh.cpu(), rtol=2e-6, atol=2e-7)
    # second producer: nearest-x2 upsample folded (8x16 -> 16x32 output), 128 couts; concat (y2 | y1) = 192 channels, 32 groups of 6
    xs = _rand((Nn, 64, 8, 16), 12, dev)
    y2, s2 = ops.conv2d(xs, ops.pack_conv_weight(w2), b2, 3, up=True, wino=ops.pack_winograd_weight(w2), stats=True)
    assert s2 is not None and tuple(s2.shape) == (Nn, 128, 4, 2)
    g2, be2 = _rand((192,), 13, dev), _rand((192,), 14, dev)
    sc, sh = ops.groupnorm_finalize(s2, g2, be2, 32, 1e-5, 16 * 32, st2=s1)
    rsc, rsh = ops.groupnorm_stats(y2, g2, be2, 32, 1e-5, x2=y1)
    assert torch.allclose(sc.cpu(), rsc.cpu(), rtol=2e-6, atol=1e-7) and torch.allclose(sh.cpu(), rsh.cpu(), rtol=2e-6, atol=2e-7)
    # a kernel without the epilogue reports 0 tiles
    _, none = ops.conv2d(x, ops.pack_conv_weight(w1[:, :, :1, :1].contiguous()), b1, 1, pad_lo=0, stats=True)
    assert none is None


# ---- round 5: conv_wino5_kernel (128-cout workgroup tiles; every wave MFMA + staging, the two halves in antiphase) -------------------
V5_CASES = [
    # (N, C1, C2, H, W, Cout, up, gn, act, temb, res)
    (3, 64, 0, 8, 16, 128, 0, 1, 1, 1, 1),     # one tile per sample, all epilogue terms
    (2, 32, 32, 24, 48, 128, 0, 1, 1, 1, 1),   # virtual concat, interior + border tiles, residual; 18 tiles on 3 persistent blocks
    (2, 32, 0, 8, 8, 128, 1, 1, 1, 1, 0),      # nearest-x2 folded, GroupNorm + SiLU
    (1, 64, 0, 8, 16, 256, 1, 0, 0, 1, 0),     # nearest-x2 folded, no GroupNorm, two cout tiles
    (3, 96, 0, 16, 32, 256, 0, 1, 0, 0, 1),    # 12 chunks, two cout tiles, no activation
    (5, 128, 0, 8, 16, 128, 0, 1, 1, 0, 0),    # several tiles per block, 16 chunks
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", V5_CASES, ids=[str(i) for i in range(len(V5_CASES))])
def test_conv_wino5_against_torch_and_bit_for_bit_against_v4(backend, case):
    """conv_wino5_kernel vs the torch fp32 convolution (1e-4) AND bit for bit against conv_wino4_kernel on the same filter image: v5
    changes who transforms what and when (one transform per 128 couts, staging spread over all eight waves, the two halves of the
    workgroup in antiphase), not one addition of the arithmetic — outputs and the GroupNorm partial sums of the epilogue must be identical.
    The launcher relies on this: it picks v4 or v5 by how many tiles a launch has, i.e. by the batch."""
    dev = select(backend)
    from audiodiffusion import _native, ops
    lib = _native.lib()
    Nn, C1, C2, H, W, Cout, up, use_gn, act, use_temb, use_res = case
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
    wp, wu = ops.pack_conv_weight(w), ops.pack_winograd_weight(w)
    outs = {}
    _native.check(lib.adm_set_option(b"conv_wino", 4))
    try:
        # bit 1 (2) = v5 wherever the shape allows (1, the default, also asks that the 128-cout tiles fill the chip); bit 3 (8) = the two
        # halves of the workgroup in antiphase instead of the default interleaved schedule (staging pieces between every wave's MFMA groups)
        for v5 in (2, 10, 0):
            _native.check(lib.adm_set_option(b"wino5", v5))
            o, st = ops.conv2d(x1, wp, b, 3, x2=x2, up=bool(up), gn=gn, act=bool(act), chan_add=temb, residual=res, wino=wu, stats=True)
            assert lib.adm_last_conv_variant() == (4315 if v5 else 4314)
            outs[v5] = (o.clone(), st.clone())
    finally:
        _native.check(lib.adm_set_option(b"wino5", -1))
        _native.check(lib.adm_set_option(b"conv_wino", -1))
    c = lambda t: None if t is None else t.cpu()  # noqa: E731
    ref = _conv_ref(c(x1), c(x2), c(w), c(b), 3, 1, up, (c(gamma), c(beta)) if use_gn else None, act, c(temb), c(res))
    assert _relerr(outs[2][0], ref) < 1e-4, _relerr(outs[2][0], ref)
    for v5 in (2, 10):
        assert torch.equal(outs[v5][0], outs[0][0]), (v5, (outs[v5][0] - outs[0][0]).abs().max())
        assert torch.equal(outs[v5][1], outs[0][1]), v5


# ---- round 5: conv_wino6_kernel — Winograd F(4x4,3x3) -----------------------------------------------------------------------------------
V6_CASES = [
    # (N, C1, C2, H, W, Cout, up, gn, act, temb, res)
    (1, 32, 0, 16, 16, 128, 0, 0, 0, 0, 0),     # one tile, nothing fused
    (2, 32, 0, 16, 32, 128, 0, 1, 1, 1, 1),     # two tiles per sample, all epilogue terms
    (3, 32, 32, 32, 16, 128, 0, 1, 1, 1, 1),    # virtual concat; 6 tiles on 3 persistent blocks (the rings roll over tile boundaries)
    (2, 64, 0, 8, 8, 128, 1, 1, 1, 1, 0),       # nearest-x2 folded (8x8 -> 16x16), GroupNorm + SiLU
    (1, 64, 0, 8, 16, 256, 1, 0, 0, 1, 0),      # nearest-x2 folded, two cout tiles, no GroupNorm
    (2, 96, 0, 32, 32, 256, 0, 1, 0, 0, 1),     # 12 chunks, interior + border tiles, two cout tiles
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", V6_CASES, ids=[str(i) for i in range(len(V6_CASES))])
def test_conv_wino6_f4x4_against_torch(backend, case):
    """conv_wino6_kernel (Winograd F(4x4,3x3): 36 points per 16 outputs, 1.78x fewer MFMAs than F(2x2,3x3)) against the torch fp32
    convolution at the per-layer bar (1e-4 of max|ref|; measured ~2-5e-6: the F(4x4) transforms cost a decimal digit against F(2x2)'s
    5e-7), and the GroupNorm partial sums of its epilogue (one (sum, sum of squares) per 16x16-pixel tile) against the stored output."""
    dev = select(backend)
    from audiodiffusion import _native, ops
    lib = _native.lib()
    Nn, C1, C2, H, W, Cout, up, use_gn, act, use_temb, use_res = case
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
    wp, wu = ops.pack_conv_weight(w), ops.pack_winograd_weight(w)
    assert wu.numel() == Cout * Ct * 52                      # the F(2x2) image and the F(4x4) image behind it
    _native.check(lib.adm_set_option(b"wino6", 2))           # 2: no plane-size floor (the default asks for planes of >= 64x64 pixels with >= 32 workgroups per sample)
    try:
        out, st = ops.conv2d(x1, wp, b, 3, x2=x2, up=bool(up), gn=gn, act=bool(act), chan_add=temb, residual=res, wino=wu, stats=True)
        assert lib.adm_last_conv_variant() == 4316
        _native.check(lib.adm_set_option(b"wino6", 0))
        out2, _ = ops.conv2d(x1, wp, b, 3, x2=x2, up=bool(up), gn=gn, act=bool(act), chan_add=temb, residual=res, wino=wu, stats=True)
        assert lib.adm_last_conv_variant() in (4314, 4315)   # the F(2x2) kernels read the first image of the same buffer
    finally:
        _native.check(lib.adm_set_option(b"wino6", -1))
    c = lambda t: None if t is None else t.cpu()  # noqa: E731
    ref = _conv_ref(c(x1), c(x2), c(w), c(b), 3, 1, up, (c(gamma), c(beta)) if use_gn else None, act, c(temb), c(res))
    assert _relerr(out, ref) < 1e-4, _relerr(out, ref)
    assert _relerr(out2, ref) < 1e-4 and _relerr(out, out2) < 1e-4
    assert tuple(st.shape) == (Nn, Cout, (Ho // 16) * (Wo // 16), 2) and not torch.isnan(st).any()
    y = out.double().reshape(Nn, Cout, Ho // 16, 16, Wo // 16, 16)
    want = torch.stack([y.sum((3, 5)).reshape(Nn, Cout, -1), (y ** 2).sum((3, 5)).reshape(Nn, Cout, -1)], -1)
    assert torch.allclose(st.cpu(), want.cpu(), rtol=2e-6, atol=2e-5)


@pytest.mark.parametrize("backend", BACKENDS)
def test_conv_wino6_is_chosen_by_the_layer_alone_and_rows_do_not_depend_on_the_batch(backend):
    """F(4x4) and F(2x2) are different arithmetic, so which of them a layer runs on must not depend on the batch (a random-weight sampler
    amplifies one bit to another picture): the default rule reads the plane size and the channel counts (>= 64x64 pixels, >= 32 tiles per sample). Row r of a batch is
    bit-identical to the sample convolved alone; a 64x64 plane of the same layer stays on the F(2x2) kernels at every batch size."""
    dev = select(backend)
    from audiodiffusion import _native, ops
    lib = _native.lib()
    w = _rand((128, 32, 3, 3), 3, dev, scale=(32 * 9) ** -0.5)
    b = _rand((128,), 4, dev)
    wp, wu = ops.pack_conv_weight(w), ops.pack_winograd_weight(w)
    x = _rand((3, 32, 128, 128), 1, dev)
    out = ops.conv2d(x, wp, b, 3, wino=wu)
    assert lib.adm_last_conv_variant() == 4316
    for r in (0, 2):
        alone = ops.conv2d(x[r:r + 1].contiguous(), wp, b, 3, wino=wu)
        assert lib.adm_last_conv_variant() == 4316 and torch.equal(alone[0], out[r])
    for n in (1, 3):
        ops.conv2d(_rand((n, 32, 64, 64), 2, dev), wp, b, 3, wino=wu)
        assert lib.adm_last_conv_variant() in (4314, 4315)


@pytest.mark.parametrize("backend", BACKENDS)
def test_a_winograd_image_packed_while_the_kernels_are_switched_off_is_the_image_they_read(backend):
    """`adm_pack_winograd_weight` under "conv_wino" = 0 used to write the 16-float layout of the experiments builds' older kernels: a
    convolution launched under the default afterwards read it in the wrong layout — and, for layers with an F(4x4) image, past its end
    (found by tools/accuracy_probe.py as a memory fault on the MI355X). The layout is now the same under 0 and 4."""
    dev = select(backend)
    from audiodiffusion import _native, ops
    lib = _native.lib()
    x1, x2 = _rand((1, 32, 16, 16), 1, dev), _rand((1, 32, 16, 16), 2, dev)
    w = _rand((128, 64, 3, 3), 3, dev, scale=(64 * 9) ** -0.5)
    b = _rand((128,), 4, dev)
    gamma, beta = _rand((64,), 5, dev), _rand((64,), 6, dev)
    gn = ops.groupnorm_stats(x1, gamma, beta, 32, 1e-5, x2=x2)
    c = lambda t: t.cpu()  # noqa: E731
    ref = _conv_ref(c(x1), c(x2), c(w), c(b), 3, 1, 0, (c(gamma), c(beta)), 1, None, None)
    try:
        _native.check(lib.adm_set_option(b"conv_wino", 0))
        wp, wu = ops.pack_conv_weight(w), ops.pack_winograd_weight(w)
        assert wu.numel() == 128 * 64 * 52
        _native.check(lib.adm_set_option(b"conv_wino", 4))
        for v6, want in ((2, 4316), (0, 4314)):
            _native.check(lib.adm_set_option(b"wino6", v6))
            out = ops.conv2d(x1, wp, b, 3, x2=x2, gn=gn, act=True, wino=wu)
            assert lib.adm_last_conv_variant() in (want, 4315)
            assert _relerr(out, ref) < 1e-4
    finally:
        _native.check(lib.adm_set_option(b"conv_wino", -1))
        _native.check(lib.adm_set_option(b"wino6", -1))


@pytest.mark.parametrize("backend", BACKENDS)
def test_the_f4x4_plane_floor_is_an_option(backend):
    """"wino6" = n >= 16 moves the plane-size floor of conv_wino6_kernel (64: the throughput setting, 256: the latency setting):
    still a function of the layer only."""
    dev = select(backend)
    from audiodiffusion import _native, ops
    lib = _native.lib()
    w = _rand((128, 32, 3, 3), 3, dev, scale=(32 * 9) ** -0.5)
    b = _rand((128,), 4, dev)
    wp, wu = ops.pack_conv_weight(w), ops.pack_winograd_weight(w)
    assert lib.adm_set_option(b"wino6", 7) != 0 and b"wino6" in lib.adm_last_error()      # neither a mode nor a floor
    try:
        for floor_px, plane, f4 in ((64, 64, True), (64, 32, False), (256, 128, False), (-1, 128, True), (-1, 64, False)):
            _native.check(lib.adm_set_option(b"wino6", floor_px))
            for n in (1, 2):
                ops.conv2d(_rand((n, 32, plane, plane), 1, dev), wp, b, 3, wino=wu)
                assert (lib.adm_last_conv_variant() == 4316) == f4, (floor_px, plane, n, lib.adm_last_conv_variant())
    finally:
        _native.check(lib.adm_set_option(b"wino6", -1))


@pytest.mark.parametrize("backend", BACKENDS)
def test_the_package_level_option_setter(backend):
    select(backend)
    import audiodiffusion
    from audiodiffusion import _native
    audiodiffusion.set_option("wino6", 256)
    try:
        with pytest.raises(_native.NativeError):
            audiodiffusion.set_option("wino6", 5)
        with pytest.raises(_native.NativeError):
            audiodiffusion.set_option("no_such_option", 1)
    finally:
        audiodiffusion.set_option("wino6", -1)


KSPLIT_CASES = [
    # ((N, C1, C2, H, W, Cout, up, gn, act, temb, res), parts): parts = what the layer rule gives (a function of the layer alone)
    ((1, 64, 0, 16, 16, 64, 0, 1, 1, 1, 1), 2),        # 8 chunks: two parts of four
    ((2, 128, 128, 16, 32, 128, 0, 1, 1, 0, 1), 8),    # virtual concat: parts 0-3 read x1, 4-7 x2; two cout tiles; residual
    ((1, 128, 0, 8, 8, 64, 1, 0, 0, 1, 0), 4),         # nearest-x2 upsample folded into the load path
    ((3, 256, 0, 32, 32, 128, 0, 1, 1, 1, 0), 8),      # 16 pixel tiles x 2 cout tiles x 8 parts x 3 samples on the emulator's 3-block grid
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case,parts", KSPLIT_CASES, ids=lambda c: "-".join(str(v) for v in c) if isinstance(c, tuple) else str(c))
def test_conv_winograd_split_k(backend, case, parts):
    """"single_sample" = 1 (the single-sample rule): conv_wino4_kernel splits the input channels of a layer whose tiles cannot fill the chip with one
    sample over `parts` workgroups per tile + one finish launch. Against torch fp32 at the usual bar, against the unsplit kernel to
    rounding, and — what the rule exists for — rows of a batch bit-identical to the same samples convolved alone."""
    dev = select(backend)
    from audiodiffusion import _native, ops
    lib = _native.lib()
    Nn, C1, C2, H, W, Cout, up, use_gn, act, use_temb, use_res = case
    x1 = _rand((Nn, C1, H, W), 1, dev)
    x2 = _rand((Nn, C2, H, W), 2, dev) if C2 else None
    Ct = C1 + C2
    w = _rand((Cout, Ct, 3, 3), 3, dev, scale=(Ct * 9) ** -0.5)
    b = _rand((Cout,), 4, dev)
    gamma, beta = _rand((Ct,), 5, dev), _rand((Ct,), 6, dev)
    temb = _rand((Nn, Cout), 7, dev) if use_temb else None
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    res = _rand((Nn, Cout, Ho, Wo), 8, dev) if use_res else None
    wp, wu = ops.pack_conv_weight(w), ops.pack_winograd_weight(w)

    def conv(rows, stats=False):
        sl = lambda t: None if t is None else t[rows].contiguous()  # noqa: E731
        gn = ops.groupnorm_stats(sl(x1), gamma, beta, 32, 1e-5, x2=sl(x2)) if use_gn else None
        return ops.conv2d(sl(x1), wp, b, 3, x2=sl(x2), up=bool(up), gn=gn, act=bool(act), chan_add=sl(temb), residual=sl(res), wino=wu,
                          stats=stats)
    everything = slice(0, Nn)
    _native.check(lib.adm_set_option(b"wino6", 0))
    try:
        unsplit = conv(everything)
        assert lib.adm_last_conv_variant() in (4314, 4315)
        _native.check(lib.adm_set_option(b"single_sample", 1))
        out, st = conv(everything, stats=True)
        assert lib.adm_last_conv_variant() == 4317, "the split-K launch was not selected"
        alone = [conv(slice(i, i + 1)) for i in range(Nn)]
    finally:
        _native.check(lib.adm_set_option(b"single_sample", -1))
        _native.check(lib.adm_set_option(b"wino6", -1))
    c = lambda t: None if t is None else t.cpu()  # noqa: E731
    ref = _conv_ref(c(x1), c(x2), c(w), c(b), 3, 1, up, (c(gamma), c(beta)) if use_gn else None, act, c(temb), c(res))
    assert _relerr(out, ref) < 1e-4, _relerr(out, ref)
    assert _relerr(out, unsplit) < 3e-6 and (parts == 1) == bool(torch.equal(out.cpu(), unsplit.cpu()))
    for i in range(Nn):
        assert torch.equal(out[i:i + 1].cpu(), alone[i].cpu()), "a sample's bits depend on its batch"
    # the finish pass leaves the GroupNorm partial sums the unsplit kernel's epilogue would: one pair per 256-pixel strip
    HW = Ho * Wo
    assert st is not None and tuple(st.shape) == (Nn, Cout, HW // 256, 2)
    y = out.double().reshape(Nn, Cout, HW // 256, 256).cpu()
    want = torch.stack([y.sum(-1), (y * y).sum(-1)], -1)
    assert torch.allclose(st.cpu(), want, rtol=2e-6, atol=2e-5)


@pytest.mark.parametrize("backend", BACKENDS)
def test_split_k_rule_is_a_function_of_the_layer(backend):
    """Which layers split, and into how many parts: planes whose 64-cout x 8x16-pixel tiles give one sample >= 256 workgroups do not; the F(4x4)
    kernel's layers do not; a channel count whose chunks do not divide into parts of four keeps fewer parts (or none); the batch never enters."""
    dev = select(backend)
    from audiodiffusion import _native, ops
    lib = _native.lib()

    def variant(N, Cin, Cout, H, W):
        x = _rand((N, Cin, H, W), 1, dev)
        w = _rand((Cout, Cin, 3, 3), 2, dev, scale=(Cin * 9) ** -0.5)
        ops.conv2d(x, ops.pack_conv_weight(w), None, 3, wino=ops.pack_winograd_weight(w))
        return lib.adm_last_conv_variant()
    _native.check(lib.adm_set_option(b"single_sample", 1))
    try:
        assert variant(1, 32, 64, 16, 16) in (4314, 4315)      # four chunks: nothing to split
        assert variant(1, 96, 64, 16, 16) in (4314, 4315)      # twelve chunks: no power-of-two partition into multiples of four
        assert variant(1, 64, 64, 16, 16) == 4317 and variant(5, 64, 64, 16, 16) == 4317      # the batch does not enter
        if backend == "hip":                                   # (too large for the emulator's fibers)
            assert variant(1, 64, 128, 128, 128) == 4316       # a layer the F(4x4) kernel takes is never split
            _native.check(lib.adm_set_option(b"wino6", 0))
            assert variant(1, 64, 128, 128, 128) in (4314, 4315)   # 8 x 16 x 2 = 256 workgroups for one sample: filled, not split
            assert variant(1, 64, 128, 64, 64) == 4317
    finally:
        _native.check(lib.adm_set_option(b"single_sample", -1))
        _native.check(lib.adm_set_option(b"wino6", -1))
    assert variant(1, 64, 64, 16, 16) in (4314, 4315)          # off by default


@pytest.mark.parametrize("backend", BACKENDS)
def test_single_sample_rule_of_the_small_plane_split(backend):
    """Part (b) of "single_sample": the split-K 3x3 kernel of the <= 8x8-pixel planes takes 16 parts (512 -> 64 channels: 64 chunks, four per part)
    instead of 8 / 4. Same kernel variant, another summation order: matches torch, agrees with the default partition to rounding, and a sample
    convolved alone has the bits of its row in a batch."""
    dev = select(backend)
    from audiodiffusion import _native, ops
    lib = _native.lib()
    x = _rand((3, 512, 8, 8), 1, dev)
    w = _rand((64, 512, 3, 3), 2, dev, scale=(512 * 9) ** -0.5)
    b = _rand((64,), 3, dev)
    wp = ops.pack_conv_weight(w)
    base = ops.conv2d(x, wp, b, 3)
    assert lib.adm_last_conv_variant() == 2316
    _native.check(lib.adm_set_option(b"single_sample", 1))
    try:
        out = ops.conv2d(x, wp, b, 3)
        assert lib.adm_last_conv_variant() == 2316
        alone = ops.conv2d(x[1:2].contiguous(), wp, b, 3)
    finally:
        _native.check(lib.adm_set_option(b"single_sample", -1))
    ref = torch.nn.functional.conv2d(x.cpu(), w.cpu(), b.cpu(), padding=1)
    assert _relerr(out, ref) < 1e-4 and _relerr(out, base) < 3e-6
    assert not torch.equal(out.cpu(), base.cpu()) and torch.equal(out[1:2].cpu(), alone.cpu())
    assert lib.adm_set_option(b"single_sample", 2) != 0 and b"single_sample" in lib.adm_last_error()


@pytest.mark.parametrize("backend", BACKENDS)
def test_single_sample_rule_of_the_stride_2_and_conv_out_layers(backend):
    """Parts (c) / (d) of "single_sample": a stride-2 3x3 convolution (Downsample2D) with a plane above 8x8 pixels splits its channels in the
    generic MFMA kernel (variant + 5), and the conv_out class kernel (Cout <= 4) splits them on planes of any size instead of walking them
    in one wide tile. Both: torch parity, agreement with the default path to rounding, a sample alone = its row in a batch."""
    dev = select(backend)
    from audiodiffusion import _native, ops
    lib = _native.lib()
    x = _rand((3, 64, 32, 32), 1, dev)
    w = _rand((64, 64, 3, 3), 2, dev, scale=(64 * 9) ** -0.5)
    b = _rand((64,), 3, dev)
    wp = ops.pack_conv_weight(w)
    xo = _rand((2, 64, 16, 64), 4, dev)
    wo = _rand((1, 64, 3, 3), 5, dev, scale=(64 * 9) ** -0.5)
    bo = _rand((1,), 6, dev)
    gamma, beta = _rand((64,), 7, dev), _rand((64,), 8, dev)
    wop = ops.pack_conv_weight(wo)

    def run(xs, xos):
        down = ops.conv2d(xs, wp, b, 3, stride=2, pad_lo=1)
        vd = lib.adm_last_conv_variant()
        gn = ops.groupnorm_stats(xos, gamma, beta, 32, 1e-5)
        out = ops.conv2d(xos, wop, bo, 3, gn=gn, act=True)
        return down, vd, out, lib.adm_last_conv_variant()
    d0, vd0, o0, vo0 = run(x, xo)
    _native.check(lib.adm_set_option(b"single_sample", 1))
    try:
        d1, vd1, o1, vo1 = run(x, xo)
        d1a, _, o1a, _ = run(x[2:3].contiguous(), xo[1:2].contiguous())
    finally:
        _native.check(lib.adm_set_option(b"single_sample", -1))
    assert vd0 in (321, 322, 324) and vd1 in (326, 327, 329), (vd0, vd1)      # 3x3 stride 2, cout tile 32 / 64 / 128: unsplit, split (+ 5)
    assert vo0 == 1002 and vo1 == 1002
    refd = torch.nn.functional.conv2d(x.cpu(), w.cpu(), b.cpu(), stride=2, padding=1)
    h = torch.nn.functional.silu(torch.nn.functional.group_norm(xo.cpu(), 32, gamma.cpu(), beta.cpu(), 1e-5))
    refo = torch.nn.functional.conv2d(h, wo.cpu(), bo.cpu(), padding=1)
    assert _relerr(d1, refd) < 1e-4 and _relerr(o1, refo) < 1e-4
    assert _relerr(d1, d0) < 3e-6 and _relerr(o1, o0) < 3e-6
    assert not torch.equal(d1.cpu(), d0.cpu()) and not torch.equal(o1.cpu(), o0.cpu())      # the rule did change the partition
    assert torch.equal(d1[2:3].cpu(), d1a.cpu()) and torch.equal(o1[1:2].cpu(), o1a.cpu())
