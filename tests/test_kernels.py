"""Per-kernel parity: HIP kernels (through the C-ABI) vs a plain torch fp32 reference of the same op.

Every test runs twice: `emu` (same kernel sources on the CPU fiber emulator, runs here) and `hip`
(the gfx950 library on a real MI355X, `-m gpu`). Tolerances follow SURVEY.md §8(c): elementwise <= 1e-6,
per-layer <= 1e-4 * max|ref|."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from native_backend import BACKENDS, select


def _rand(shape, seed, dev, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


def _relerr(a, b):
    return float((a.cpu() - b.cpu()).abs().max() / (b.cpu().abs().max() + 1e-30))


@pytest.mark.parametrize("backend", BACKENDS)
def test_sched_step_ddim_ddpm_all_timesteps(backend):
    dev = select(backend)
    from audiodiffusion import ops, schedulers
    from oracle.schedulers import DDIMScheduler as ODDIM, DDPMScheduler as ODDPM
    x, e, nz = (_rand((2, 1, 8, 16), s, dev) for s in (1, 2, 3))
    for kind, steps, eta in (("ddim", 50, 0.0), ("ddim", 50, 0.7), ("ddpm", 1000, 0.0), ("ddpm", 10, 0.0)):
        ref = ODDIM() if kind == "ddim" else ODDPM()
        mine = schedulers.DDIMScheduler() if kind == "ddim" else schedulers.DDPMScheduler()
        ref.set_timesteps(steps), mine.set_timesteps(steps)
        assert mine.timesteps.tolist() == ref.timesteps.tolist()
        table = mine.coef_table(dev, eta)
        idx = range(steps) if steps <= 50 else list(range(0, 1000, 37)) + [998, 999]
        for i in idx:
            t = int(mine.timesteps[i])
            if kind == "ddim":
                r = ref.step(e.cpu(), t, x.cpu(), eta=eta, variance_noise=nz.cpu())["prev_sample"]
            else:
                r = ref.step(e.cpu(), t, x.cpu(), variance_noise=nz.cpu())["prev_sample"]
            o = ops.sched_step(x, e, table, i, noise=nz)
            assert float((o.cpu() - r).abs().max()) <= 2e-6 * max(1.0, float(r.abs().max())), (kind, steps, i)
        # the drop-in .step() member (what pipeline_audio_diffusion.py:166-179 calls)
        t = mine.timesteps[3]
        kw = dict(eta=eta) if kind == "ddim" else {}
        o = mine.step(model_output=e, timestep=t, sample=x, variance_noise=nz, **kw)["prev_sample"]
        r = ref.step(e.cpu(), int(t), x.cpu(), variance_noise=nz.cpu(), **kw)["prev_sample"]
        assert float((o.cpu() - r).abs().max()) <= 2e-6 * max(1.0, float(r.abs().max()))


@pytest.mark.parametrize("backend", BACKENDS)
def test_sched_step_mask_and_u8(backend):
    dev = select(backend)
    from audiodiffusion import ops
    B, H, W, n = 2, 4, 16, 3
    x, e = _rand((B, 1, H, W), 1, dev), _rand((B, 1, H, W), 2, dev)
    mask = _rand((B, n, H, W), 3, dev)
    row = dict(sqrt_beta=0.3, sqrt_alpha=0.9, clip=1.0, k_x0=0.8, k_x=0.0, k_eps=0.1, k_noise=0.0, timestep=0)
    table = ops.sched_coef_table([row] * n, dev)
    u8 = torch.zeros((B, H, W), dtype=torch.uint8, device=dev)
    o = ops.sched_step(x, e, table, 1, mask=mask, mask_start=3, mask_end=5, u8_out=u8)
    x0 = ((x - 0.3 * e) / 0.9).clamp(-1, 1)
    r = 0.8 * x0 + 0.1 * e
    r[:, :, :, :3] = mask[:, 1:2, :, :3]
    r[:, :, :, -5:] = mask[:, 1:2, :, -5:]
    assert torch.allclose(o.cpu(), r.cpu(), atol=1e-6)
    q = ((o.cpu() / 2 + 0.5).clamp(0, 1).numpy() * 255).round().astype("uint8")[:, 0]
    assert np.array_equal(u8.cpu().numpy(), q)
    # dequant rule on exact ties: 0.5 -> 0 (half-to-even), 127.5 -> 128
    ties = torch.tensor([(0.5 / 255 - 0.5) * 2, (127.5 / 255 - 0.5) * 2, -3.0, 3.0], device=dev)
    assert ops.dequant_u8(ties.contiguous()).cpu().tolist() == [0, 128, 0, 255]


@pytest.mark.parametrize("backend", BACKENDS)
def test_add_noise_modes(backend):
    dev = select(backend)
    from audiodiffusion import ops, schedulers
    from oracle.schedulers import DDPMScheduler as ODDPM
    x0, noise = _rand((1, 8, 8), 1, dev), _rand((3, 1, 8, 8), 2, dev)
    mine, ref = schedulers.DDPMScheduler(), ODDPM()
    ts = torch.tensor([999, 500, 20, 0])
    m = mine.add_noise(x0, noise, ts)                      # (B, n, H, W) mask build, pipeline:157
    r = ref.add_noise(x0.cpu(), noise.cpu(), ts)
    assert m.shape == (3, 4, 8, 8) and torch.allclose(m.cpu(), r, atol=1e-6)
    xs = _rand((3, 1, 8, 8), 4, dev)
    p = mine.add_noise(xs, noise, ts[:3])                  # per-sample timesteps, train_unet.py:250
    assert torch.allclose(p.cpu(), ref.add_noise(xs.cpu(), noise.cpu(), ts[:3]), atol=1e-6)
    one = mine.add_noise(x0, noise[:1].contiguous(), ts[1])  # pipeline:150
    assert torch.allclose(one.cpu()[0, 0], ref.add_noise(x0.cpu(), noise[:1].cpu(), ts[1])[0, 0], atol=1e-6)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("C1,C2,HW", [(32, 0, 64), (64, 32, 16), (96, 96, 4), (32, 0, 1), (64, 0, 1024)])
def test_groupnorm_stats(backend, C1, C2, HW):
    dev = select(backend)
    from audiodiffusion import ops
    h = int(round(HW ** 0.5))
    x1 = _rand((2, C1, h, HW // h), 1, dev) * 3 + 1.5
    x2 = _rand((2, C2, h, HW // h), 2, dev) if C2 else None
    gamma, beta = _rand((C1 + C2,), 3, dev), _rand((C1 + C2,), 4, dev)
    sc, sh = ops.groupnorm_stats(x1, gamma, beta, 32, 1e-5, x2=x2)
    xc = torch.cat([x1, x2], 1) if C2 else x1
    ref = F.group_norm(xc.cpu(), 32, gamma.cpu(), beta.cpu(), 1e-5)
    got = xc.cpu() * sc.cpu()[:, :, None, None] + sh.cpu()[:, :, None, None]
    # (32,0,1): one element per group -> var == 0, rstd = eps^-0.5 = 316: the x*scale+shift form (also ATen's)
    # cancels at ~1e-7*316; degenerate, only reachable with toy configs.
    assert _relerr(got, ref) < (2e-4 if HW * (C1 + C2) // 32 == 1 else 2e-5)


CONV_CASES = [
    # (N, C1, C2, H, W, Cout, ks, stride, up, gn, act, temb, res)
    (1, 32, 0, 16, 16, 32, 3, 1, 0, 1, 1, 1, 0),     # resnet conv1
    (2, 32, 0, 8, 8, 64, 3, 1, 0, 1, 1, 0, 1),       # conv2 + residual, NI=2 tiles
    (1, 32, 32, 16, 32, 32, 3, 1, 0, 1, 1, 1, 0),    # virtual concat, group straddle
    (1, 64, 0, 16, 16, 64, 3, 2, 0, 0, 0, 0, 0),     # downsample
    (3, 32, 0, 4, 4, 32, 3, 1, 1, 0, 0, 0, 0),       # upsample folded, odd batch
    (2, 64, 32, 8, 8, 32, 1, 1, 0, 0, 0, 0, 0),      # 1x1 shortcut over concat
    (2, 32, 0, 2, 2, 32, 3, 1, 0, 1, 1, 1, 1),       # 2x2 level
    (5, 32, 0, 1, 1, 64, 3, 1, 0, 1, 1, 0, 1),       # 1x1 level (latent 32x32 bottom)
    (2, 32, 0, 2, 2, 32, 3, 2, 0, 0, 0, 0, 0),       # stride 2 -> 1x1
    (1, 128, 0, 8, 16, 128, 3, 1, 0, 1, 1, 1, 1),    # production channel count
    (1, 32, 0, 8, 8, 96, 1, 1, 0, 1, 0, 0, 0),       # fused q|k|v projection (GN, no SiLU)
]


def _conv_ref(x1, x2, w, b, ks, stride, up, gn, act, temb, res):
    x = torch.cat([x1, x2], 1) if x2 is not None else x1
    if gn is not None:
        x = F.group_norm(x, 32, gn[0], gn[1], 1e-5)
    if act:
        x = F.silu(x)
    if up:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    y = F.conv2d(x, w, b, stride=stride, padding=ks // 2)
    if temb is not None:
        y = y + temb[:, :, None, None]
    if res is not None:
        y = y + res
    return y


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", CONV_CASES, ids=[str(i) for i in range(len(CONV_CASES))])
def test_conv2d_fused(backend, case):
    dev = select(backend)
    from audiodiffusion import ops
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
    out = ops.conv2d(x1, ops.pack_conv_weight(w), b, ks, x2=x2, up=bool(up), stride=stride, pad_lo=1,
                     gn=gn, act=bool(act), chan_add=temb, residual=res)
    c = lambda t: None if t is None else t.cpu()  # noqa: E731
    ref = _conv_ref(c(x1), c(x2), c(w), c(b), ks, stride, up, (c(gamma), c(beta)) if use_gn else None, act, c(temb), c(res))
    assert out.shape == ref.shape
    assert _relerr(out, ref) < 1e-4, _relerr(out, ref)


@pytest.mark.parametrize("backend", BACKENDS)
def test_conv_in_out_small(backend):
    dev = select(backend)
    from audiodiffusion import ops
    x = _rand((2, 1, 16, 32), 1, dev)
    w = _rand((32, 1, 3, 3), 2, dev, 0.3)
    b = _rand((32,), 3, dev)
    out = ops.conv2d(x, ops.pack_conv_weight(w), b, 3)
    assert _relerr(out, F.conv2d(x.cpu(), w.cpu(), b.cpu(), padding=1)) < 1e-5
    # conv_out: GN+SiLU -> 32 -> 1
    h = _rand((2, 32, 16, 32), 4, dev)
    gamma, beta = _rand((32,), 5, dev), _rand((32,), 6, dev)
    w2, b2 = _rand((1, 32, 3, 3), 7, dev, 0.1), _rand((1,), 8, dev)
    gn = ops.groupnorm_stats(h, gamma, beta, 32, 1e-5)
    out = ops.conv2d(h, ops.pack_conv_weight(w2), b2, 3, gn=gn, act=True)
    ref = F.conv2d(F.silu(F.group_norm(h.cpu(), 32, gamma.cpu(), beta.cpu(), 1e-5)), w2.cpu(), b2.cpu(), padding=1)
    assert _relerr(out, ref) < 1e-4


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("Cin,Cout,H,W,N", [(128, 64, 8, 8, 3), (256, 96, 8, 8, 5), (160, 128, 8, 8, 2)])
def test_conv3x3_split_k_at_tiny_spatial_sizes(backend, Cin, Cout, H, W, N):
    """3x3 stride-1 layers whose tiles cannot fill the chip run on 64- (or 32-) cout tiles with K split over several workgroups:
    partial slabs + ksplit_finish_kernel (bias, per-sample term, residual added there). GroupNorm + SiLU on the load path, several
    images per 128-pixel tile, a chunk count that does not divide by the split, and the residual aliasing the output in place."""
    dev = select(backend)
    from audiodiffusion import _native, ops
    x = _rand((N, Cin, H, W), 1, dev)
    w = _rand((Cout, Cin, 3, 3), 3, dev, scale=(Cin * 9) ** -0.5)
    b = _rand((Cout,), 4, dev)
    gamma, beta = _rand((Cin,), 5, dev), _rand((Cin,), 6, dev)
    gn = ops.groupnorm_stats(x, gamma, beta, 32, 1e-5)
    temb = _rand((N, Cout), 7, dev)
    res = _rand((N, Cout, H, W), 8, dev)
    wp = ops.pack_conv_weight(w)
    out = ops.conv2d(x, wp, b, 3, gn=gn, act=True, chan_add=temb, residual=res)
    assert _native.lib().adm_last_conv_variant() == 2316, "the split-K instantiation was not selected"
    c = lambda t: t.cpu()  # noqa: E731
    ref = _conv_ref(c(x), None, c(w), c(b), 3, 1, 0, (c(gamma), c(beta)), True, c(temb), c(res))
    assert _relerr(out, ref) < 1e-4, _relerr(out, ref)
    # the residual may alias the output (the backward pass accumulates data gradients in place): out = conv + out
    acc = res.clone()
    import ctypes as C
    a = ops._conv_args(x, wp, b, 3, None, False, 1, 1, gn, True, Cout)
    a.residual, a.out = _native.ptr(acc), _native.ptr(acc)
    a.chan_add, a.chan_add_stride = C.c_void_p(temb.data_ptr()), temb.stride(0)
    _native.check(_native.lib().adm_conv2d(C.byref(a), _native.stream_for(x)))
    assert _relerr(acc, ref) < 1e-4
    # adm_release_stream: the stream's split-K slab buffer is given back (a caller about to destroy the stream does this); the
    # next split-K launch on that stream takes a fresh one and produces the same bits; a stream the library holds nothing for is a no-op
    st = _native.stream_for(x)
    _native.check(_native.lib().adm_release_stream(st))
    _native.check(_native.lib().adm_release_stream(st))
    again = ops.conv2d(x, wp, b, 3, gn=gn, act=True, chan_add=temb, residual=res)
    assert _native.lib().adm_last_conv_variant() == 2316 and torch.equal(again, out)


TINY_LEVELS = [  # (N, C1, C2, H, W, Cout, ks, stride, up, use_gn, act, use_temb, use_res, expected variant)
    (16, 128, 0, 1, 1, 128, 3, 1, 0, 1, 1, 1, 1, 319),     # 1x1-pixel level: 16 of a tile's 128 columns, HW = 1 (scalar finish)
    (5, 64, 64, 2, 2, 64, 3, 1, 0, 1, 1, 1, 0, 317),       # 2x2, virtual concat, 64-cout tiles
    (3, 96, 0, 4, 4, 96, 3, 1, 0, 1, 1, 0, 1, 316),        # 4x4 (CS = 288 > the pipelined kernel's plan), 32-cout tiles
    (16, 128, 0, 1, 1, 128, 3, 1, 1, 0, 0, 0, 0, 319),     # Upsample2D 1x1 -> 2x2 folded into the load path
    (4, 128, 0, 2, 2, 128, 3, 2, 0, 0, 0, 0, 0, 329),      # Downsample2D 2x2 -> 1x1 (stride 2)
    (16, 160, 0, 2, 2, 128, 1, 1, 0, 0, 0, 0, 1, 119),     # 1x1 projection / shortcut on a 2x2 plane (+ residual)
    (130, 64, 0, 1, 1, 64, 3, 1, 0, 1, 1, 1, 1, 317),      # more images than one tile holds: two tile groups, the second ragged
    (6, 64, 0, 1, 4, 64, 3, 1, 0, 1, 1, 0, 0, 317),        # a 1 x 4 plane: only the middle ROW of taps is live (tap mask 0b000111000)
    (6, 64, 0, 4, 1, 64, 3, 1, 0, 0, 0, 1, 0, 317),        # a 4 x 1 plane: only the middle COLUMN of taps
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", TINY_LEVELS, ids=[f"{c[4]}x{c[3]}-k{c[6]}s{c[7]}u{c[8]}-n{c[0]}" for c in TINY_LEVELS])
def test_conv_split_k_on_the_generic_kernel_at_the_deepest_levels(backend, case):
    """The 4x4 / 2x2 / 1x1-pixel levels of a 64x64 or latent 32x32 model (BASELINE configs 1 and 4): the generic kernel with K split
    over up to 64 workgroups per tile + the slab reduction (float4 and scalar forms), every fusion of the load path and epilogue."""
    dev = select(backend)
    from audiodiffusion import _native, ops
    Nn, C1, C2, H, W, Cout, ks, stride, up, use_gn, act, use_temb, use_res, want = case
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
    out = ops.conv2d(x1, ops.pack_conv_weight(w), b, ks, x2=x2, up=bool(up), stride=stride, pad_lo=1,
                     gn=gn, act=bool(act), chan_add=temb, residual=res)
    assert _native.lib().adm_last_conv_variant() == want, _native.lib().adm_last_conv_variant()
    c = lambda t: None if t is None else t.cpu()  # noqa: E731
    ref = _conv_ref(c(x1), c(x2), c(w), c(b), ks, stride, up, (c(gamma), c(beta)) if use_gn else None, act, c(temb), c(res))
    assert out.shape == ref.shape
    assert _relerr(out, ref) < 1e-4, _relerr(out, ref)
    again = ops.conv2d(x1, ops.pack_conv_weight(w), b, ks, x2=x2, up=bool(up), stride=stride, pad_lo=1,
                       gn=gn, act=bool(act), chan_add=temb, residual=res)
    assert torch.equal(out, again), "the slab reduction must be deterministic"


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("Cin,Cout,H,W,ks,stride", [(128, 128, 1, 1, 3, 1), (128, 64, 4, 4, 3, 1), (256, 64, 8, 8, 3, 1), (128, 128, 8, 8, 3, 2),
                                                    (128, 64, 2, 2, 1, 1), (64, 64, 16, 16, 3, 1)])
def test_a_samples_convolution_does_not_depend_on_the_batch_it_is_in(backend, Cin, Cout, H, W, ks, stride):
    """Row r of a 70-sample launch == the same sample convolved alone, BIT FOR BIT: whether K is split, and into how many parts, is a
    function of the layer, not of the batch (a split chosen from the tile count would change the fp32 summation order with the
    batch size — and with it what a random-weight sampler makes of a row on 1 GPU vs 8)."""
    dev = select(backend)
    from audiodiffusion import ops
    x = _rand((70, Cin, H, W), 1, dev)
    w = _rand((Cout, Cin, ks, ks), 3, dev, scale=(Cin * ks * ks) ** -0.5)
    wp, b = ops.pack_conv_weight(w), _rand((Cout,), 4, dev)
    full = ops.conv2d(x, wp, b, ks, stride=stride)
    for r in (0, 33, 69):
        one = ops.conv2d(x[r:r + 1].contiguous(), wp, b, ks, stride=stride)
        assert torch.equal(one[0], full[r]), r


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("Cin,Cout,H,W,N", [(128, 1, 32, 32, 3), (64, 2, 16, 24, 5), (72, 1, 8, 8, 2)])
def test_conv_in_out_class_on_small_images(backend, Cin, Cout, H, W, N):
    """A latent / 64x64 model's first and last layers: conv_out's channel loop split over 8 workgroups per tile (+ the slab
    reduction with bias and residual), conv_in's output channels split over the grid; rows do not depend on the batch."""
    dev = select(backend)
    from audiodiffusion import ops
    h = _rand((N, Cin, H, W), 4, dev)
    gamma, beta = _rand((Cin,), 5, dev), _rand((Cin,), 6, dev)
    w2, b2 = _rand((Cout, Cin, 3, 3), 7, dev, 0.1), _rand((Cout,), 8, dev)
    res = _rand((N, Cout, H, W), 9, dev)
    groups = 32 if Cin % 32 == 0 else 8
    gn = ops.groupnorm_stats(h, gamma, beta, groups, 1e-5)
    out = ops.conv2d(h, ops.pack_conv_weight(w2), b2, 3, gn=gn, act=True, residual=res)
    ref = F.conv2d(F.silu(F.group_norm(h.cpu(), groups, gamma.cpu(), beta.cpu(), 1e-5)), w2.cpu(), b2.cpu(), padding=1) + res.cpu()
    assert _relerr(out, ref) < 1e-4, _relerr(out, ref)
    gn1 = (gn[0][1:2].contiguous(), gn[1][1:2].contiguous())
    one = ops.conv2d(h[1:2].contiguous(), ops.pack_conv_weight(w2), b2, 3, gn=gn1, act=True, residual=res[1:2].contiguous())
    assert torch.equal(one[0], out[1])
    x = _rand((N, Cout, H, W), 1, dev)                    # conv_in class: Cout (<= 2) input channels -> 128
    w1, b1 = _rand((128, Cout, 3, 3), 2, dev, 0.3), _rand((128,), 3, dev)
    o1 = ops.conv2d(x, ops.pack_conv_weight(w1), b1, 3)
    assert _relerr(o1, F.conv2d(x.cpu(), w1.cpu(), b1.cpu(), padding=1)) < 1e-5


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("Cin,H,W", [(1, 16, 32), (1, 40, 52), (3, 8, 12)])
def test_conv_in_statistics_epilogue(backend, Cin, H, W):
    """conv_in class, four pixels per thread: the output, and the GroupNorm partial sums its epilogue writes per (sample, cout,
    wave of 256 pixels) — adm_groupnorm_finalize on them must give what the read pass (adm_groupnorm_stats) gives on the output
    (ragged last wave: 40 x 52 = 2080 pixels = 8 full waves + 32 pixels; 8 x 12: a single partial wave)."""
    dev = select(backend)
    from audiodiffusion import ops
    x = _rand((2, Cin, H, W), 1, dev)
    w = _rand((64, Cin, 3, 3), 2, dev, 0.3)
    b = _rand((64,), 3, dev)
    out, st = ops.conv2d(x, ops.pack_conv_weight(w), b, 3, stats=True)
    assert _relerr(out, F.conv2d(x.cpu(), w.cpu(), b.cpu(), padding=1)) < 1e-5
    assert st is not None and st.shape[2] == ((H * W // 4 + 255) // 256) * 4 and not torch.isnan(st).any()
    gamma, beta = _rand((64,), 5, dev), _rand((64,), 6, dev)
    sc, sh = ops.groupnorm_finalize(st, gamma, beta, 32, 1e-5, H * W)
    rsc, rsh = ops.groupnorm_stats(out, gamma, beta, 32, 1e-5)
    assert _relerr(sc, rsc) < 1e-5 and _relerr(sh, rsh) < 1e-5


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("Cin,Cout,H,W,use_gn,use_res", [(32, 1, 32, 128, 1, 0), (12, 3, 16, 64, 0, 1), (64, 1, 48, 192, 1, 1)])
def test_conv_out_wide_tiles(backend, Cin, Cout, H, W, use_gn, use_res):
    """conv_out class on rows of whole 64-pixel tiles (conv_small_cout_wide_kernel): several tiles in both directions (left /
    inner / right halo columns, top / bottom padding rows), a channel count that is not a multiple of the 8-channel chunk,
    residual, up to 3 output channels; and a crop to a width of 48 pixels through the 16x16-tile kernel."""
    dev = select(backend)
    from audiodiffusion import ops
    h = _rand((2, Cin, H, W), 4, dev)
    w2, b2 = _rand((Cout, Cin, 3, 3), 7, dev, 0.1), _rand((Cout,), 8, dev)
    res = _rand((2, Cout, H, W), 9, dev) if use_res else None
    gn, a = None, h.cpu()
    if use_gn:
        gamma, beta = _rand((Cin,), 5, dev), _rand((Cin,), 6, dev)
        gn = ops.groupnorm_stats(h, gamma, beta, 32 if Cin % 32 == 0 else 4, 1e-5)
        a = F.silu(F.group_norm(a, 32 if Cin % 32 == 0 else 4, gamma.cpu(), beta.cpu(), 1e-5))
    out = ops.conv2d(h, ops.pack_conv_weight(w2), b2, 3, gn=gn, act=bool(use_gn), residual=res)
    ref = F.conv2d(a, w2.cpu(), b2.cpu(), padding=1) + (res.cpu() if use_res else 0)
    assert _relerr(out, ref) < 1e-4
    # same inputs through the narrow kernel: a width that is not a multiple of 64 selects it (crop of the same problem)
    hc = h[..., :48].contiguous()
    gnc = None
    if use_gn:
        gnc = ops.groupnorm_stats(hc, gamma, beta, 32 if Cin % 32 == 0 else 4, 1e-5)
        ac = F.silu(F.group_norm(hc.cpu(), 32 if Cin % 32 == 0 else 4, gamma.cpu(), beta.cpu(), 1e-5))
    else:
        ac = hc.cpu()
    outc = ops.conv2d(hc, ops.pack_conv_weight(w2), b2, 3, gn=gnc, act=bool(use_gn))
    assert _relerr(outc, F.conv2d(ac, w2.cpu(), b2.cpu(), padding=1)) < 1e-4


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("C,T,d", [(32, 64, 8), (64, 256, 8), (32, 16, 8), (32, 4, 8)])
def test_attention_core(backend, C, T, d):
    dev = select(backend)
    from audiodiffusion import ops
    h = int(round(T ** 0.5))
    qkv = _rand((2, 3 * C, h, T // h), 1, dev)
    out = ops.attention(qkv, d)
    q, k, v = qkv.cpu().view(2, 3, C // d, d, T).unbind(1)          # (N, heads, d, T)
    s = torch.einsum("nhdt,nhdj->nhtj", q, k) * d ** -0.5
    ref = torch.einsum("nhtj,nhdj->nhdt", s.softmax(-1), v).reshape(2, C, h, T // h)
    assert _relerr(out, ref) < 1e-5


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("C,T,d", [(64, 256, 8), (32, 64, 8), (16, 100, 4)], ids=["16x16", "8x8", "10x10-ragged-block"])
def test_attention_under_the_single_sample_rule(backend, C, T, d):
    """"single_sample" = 1: attention_split4_kernel — four lanes per query, each a quarter of the keys (a sample of the 256x256 model gives the
    one-lane-per-query kernel 64 workgroups with one wave per SIMD). Same softmax; the sums are combined as (p0 + p1) + (p2 + p3): torch
    parity at the kernel's bar, agreement with the default kernel to rounding, and a sample alone = its row in a batch."""
    dev = select(backend)
    from audiodiffusion import _native, ops
    h = int(round(T ** 0.5))
    qkv = _rand((3, 3 * C, h, T // h), 1, dev)
    base = ops.attention(qkv, d)
    _native.check(_native.lib().adm_set_option(b"single_sample", 1))
    try:
        out = ops.attention(qkv, d)
        alone = ops.attention(qkv[2:3].contiguous(), d)
    finally:
        _native.check(_native.lib().adm_set_option(b"single_sample", -1))
    q, k, v = qkv.cpu().view(3, 3, C // d, d, T).unbind(1)
    s = torch.einsum("nhdt,nhdj->nhtj", q, k) * d ** -0.5
    ref = torch.einsum("nhtj,nhdj->nhdt", s.softmax(-1), v).reshape(3, C, h, T // h)
    assert _relerr(out, ref) < 1e-5 and _relerr(out, base) < 1e-6
    assert not torch.equal(out.cpu(), base.cpu()) and torch.equal(out[2:3].cpu(), alone.cpu())
