"""Tensor-level wrappers over the op-level C-ABI entry points (parity tests and the scheduler shims use these).

Every function launches hand-written HIP kernels from libadm_hip.so on torch's current stream; none has a
PyTorch/CPU fallback.
"""
import ctypes as C

import torch

from . import _native as N


def _f32(t):
    assert t.dtype == torch.float32 and t.is_contiguous()
    return t


def sched_coef_table(rows, device):
    """rows: list of dicts/tuples with the 8 adm_sched_coef fields -> (n,8) fp32 device tensor."""
    t = torch.tensor([[float(r[k]) for k in ("sqrt_beta", "sqrt_alpha", "clip", "k_x0", "k_x", "k_eps", "k_noise", "timestep")]
                      for r in rows], dtype=torch.float32)
    return t.to(device)


def sched_step(x, eps, coef_table, step, noise=None, mask=None, mask_start=0, mask_end=0, out=None, u8_out=None):
    """Fused scheduler epilogue (pipeline_audio_diffusion.py:165-185,192-194). x,eps: (B,C,H,W)."""
    _f32(x), _f32(eps)
    B, Cc, H, W = x.shape
    out = torch.empty_like(x) if out is None else out
    n_mask = mask.shape[1] if mask is not None else 0
    N.check(N.lib().adm_sched_step(N.ptr(x), N.ptr(eps), N.ptr(noise), N.ptr(out), N.ptr(u8_out), N.ptr(coef_table),
                                   None, int(step), N.ptr(mask), n_mask, int(mask_start), int(mask_end), B, Cc, H, W,
                                   N.stream_for(x)))
    return out


def add_noise(x0, noise, sa, sb, per_sample):
    """scheduler.add_noise. per_sample=True: x0,noise (B,...) with sa,sb (B,) -> (B,...).
    per_sample=False (mask build, pipeline:157): x0 (1,H,W) broadcast, noise (B,1,H,W), sa,sb (n,) -> (B,n,H,W)."""
    _f32(x0), _f32(noise), _f32(sa), _f32(sb)
    B = noise.shape[0]
    P = noise[0].numel()
    if per_sample:
        out = torch.empty_like(noise)
        N.check(N.lib().adm_add_noise(N.ptr(x0), P, N.ptr(noise), N.ptr(sa), N.ptr(sb), 1, 0, N.ptr(out), B, 1, P,
                                      N.stream_for(noise)))
    else:
        n = sa.numel()
        out = torch.empty((B, n) + tuple(noise.shape[2:]), dtype=torch.float32, device=noise.device)
        N.check(N.lib().adm_add_noise(N.ptr(x0), 0, N.ptr(noise), N.ptr(sa), N.ptr(sb), 0, 1, N.ptr(out), B, n, P,
                                      N.stream_for(noise)))
    return out


def dequant_u8(x):
    """(x/2+0.5).clamp(0,1)*255 -> round-half-even -> uint8, same shape (pipeline_audio_diffusion.py:192-194)."""
    _f32(x)
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    N.check(N.lib().adm_dequant_u8(N.ptr(x), N.ptr(out), x.numel(), N.stream_for(x)))
    return out


def slerp_grid(x0, x1, alphas):
    """AudioDiffusionPipeline.slerp for every alpha in one launch pair: (len(alphas),) + x0.shape."""
    _f32(x0), _f32(x1)
    assert x0.shape == x1.shape
    al = torch.as_tensor([float(a) for a in alphas], dtype=torch.float64).to(x0.device)
    out = torch.empty((al.numel(),) + tuple(x0.shape), dtype=torch.float32, device=x0.device)
    scratch = torch.zeros(3, dtype=torch.float64, device=x0.device)
    N.check(N.lib().adm_slerp_grid(N.ptr(x0.contiguous()), N.ptr(x1.contiguous()), x0.numel(), N.ptr(al), al.numel(), N.ptr(out),
                                   N.ptr(scratch), N.stream_for(x0)))
    return out


def groupnorm_stats(x1, gamma, beta, groups, eps, x2=None):
    """Returns per-(n,c) (scale, shift) of GroupNorm over the virtual concat (x1|x2)."""
    _f32(x1)
    Nn, C1 = x1.shape[:2]
    C2 = x2.shape[1] if x2 is not None else 0
    HW = x1[0, 0].numel()
    scale = torch.empty((Nn, C1 + C2), dtype=torch.float32, device=x1.device)
    shift = torch.empty_like(scale)
    N.check(N.lib().adm_groupnorm_stats(N.ptr(x1), C1, N.ptr(x2), C2, Nn, HW, groups, float(eps), N.ptr(gamma),
                                        N.ptr(beta), N.ptr(scale), N.ptr(shift), N.stream_for(x1)))
    return scale, shift


def pack_conv_weight(w):
    """(Cout,Cin,ks,ks) -> [Cin][ks*ks][Cout]."""
    _f32(w)
    co, ci, ks, _ = w.shape
    wp = torch.empty((ci, ks * ks, co), dtype=torch.float32, device=w.device)
    N.check(N.lib().adm_pack_conv_weight(N.ptr(w), N.ptr(wp), co, ci, ks, N.stream_for(w)))
    return wp


def pack_conv_weight_T(w):
    """(Cout,Cin,ks,ks) -> backward-data packing [Cout][ks*ks][Cin] (transposed + flipped)."""
    _f32(w)
    co, ci, ks, _ = w.shape
    wp = torch.empty((co, ks * ks, ci), dtype=torch.float32, device=w.device)
    N.check(N.lib().adm_pack_conv_weight_T(N.ptr(w), N.ptr(wp), co, ci, ks, N.stream_for(w)))
    return wp


def pack_winograd_weight(w):
    """(Cout,Cin,3,3) -> Winograd-domain U = G g G^T as [Cin][16][Cout]."""
    _f32(w)
    co, ci = w.shape[:2]
    # the buffer holds the F(2x2,3x3) image and, where conv_wino6_kernel may run, the F(4x4,3x3) image behind it (adm.h)
    wu = torch.empty((int(N.lib().adm_winograd_packed_floats(co, ci, 0)),), dtype=torch.float32, device=w.device)
    N.check(N.lib().adm_pack_winograd_weight(N.ptr(w), N.ptr(wu), co, ci, N.stream_for(w)))
    return wu


def pack_winograd_weight_T(w):
    """(Cout,Cin,3,3) -> Winograd-domain filters of the data-gradient convolution as [Cout][16][Cin]."""
    _f32(w)
    co, ci = w.shape[:2]
    wu = torch.empty((int(N.lib().adm_winograd_packed_floats(co, ci, 1)),), dtype=torch.float32, device=w.device)
    N.check(N.lib().adm_pack_winograd_weight_T(N.ptr(w), N.ptr(wu), co, ci, N.stream_for(w)))
    return wu


def pack_bf16_weight(w, transposed=False):
    """(Cout,Cin,3,3) fp32 -> bf16 MFMA operand layout [tap][Cin/8][Cout][8] (transposed: the data-gradient filters)."""
    _f32(w)
    co, ci = w.shape[:2]
    ks = w.shape[2] if w.dim() == 4 else 1
    shape = (ks * ks, co // 8, ci, 8) if transposed else (ks * ks, ci // 8, co, 8)
    wb = torch.empty(shape, dtype=torch.bfloat16, device=w.device)
    N.check(N.lib().adm_pack_bf16_weight_ks(N.ptr(w), C.c_void_p(wb.data_ptr()), co, ci, ks, int(transposed), N.stream_for(w)))
    return wb


def conv2d(x1, wpacked, bias, ks, x2=None, up=False, stride=1, pad_lo=1, gn=None, act=False, chan_add=None,
           residual=None, wino=None, bf16=None, stats=False):
    """Fused convolution (see include/adm.h adm_conv_args). stats=True: also returns the GroupNorm partial sums of the output
    the kernel's epilogue wrote, (N, Cout, tiles, 2) fp64 — None when the dispatched kernel has no such epilogue."""
    _f32(x1)
    Nn, C1, H, W = x1.shape
    Cout = wpacked.shape[2]
    Ho, Wo = C.c_int(), C.c_int()
    N.lib().adm_conv_out_dims(H, W, int(up), stride, ks, pad_lo, C.byref(Ho), C.byref(Wo))
    out = torch.empty((Nn, Cout, Ho.value, Wo.value), dtype=torch.float32, device=x1.device)
    a = N.ConvArgs()
    a.x1, a.C1 = N.ptr(x1), C1
    a.x2, a.C2 = (N.ptr(x2), x2.shape[1]) if x2 is not None else (None, 0)
    a.N, a.H, a.W = Nn, H, W
    a.up, a.stride, a.ks, a.pad_lo = int(up), stride, ks, pad_lo  # up: 0 none, 1/True nearest x2, 2 zero-insertion x2
    if gn is not None:
        a.gn_scale, a.gn_shift = N.ptr(gn[0]), N.ptr(gn[1])
    a.act = int(act)
    a.wpacked, a.bias, a.Cout = N.ptr(wpacked), N.ptr(bias), Cout
    if chan_add is not None:
        assert chan_add.dtype == torch.float32 and chan_add.stride(1) == 1
        a.chan_add, a.chan_add_stride = C.c_void_p(chan_add.data_ptr()), chan_add.stride(0)
    a.residual = N.ptr(residual)
    a.wino_packed = N.ptr(wino)
    a.bf16_packed = C.c_void_p(bf16.data_ptr()) if bf16 is not None else None
    a.out = N.ptr(out)
    st = None
    if stats:
        tiles = N.lib().adm_conv_stats_tiles(C.byref(a))
        if tiles > 0:
            st = torch.full((Nn, Cout, tiles, 2), float("nan"), dtype=torch.float64, device=x1.device)
            a.stats_out, a.stats_tiles = N.ptr(st), tiles
    N.check(N.lib().adm_conv2d(C.byref(a), N.stream_for(x1)))
    return (out, st) if stats else out


def groupnorm_finalize(st1, gamma, beta, groups, eps, hw, st2=None):
    """GroupNorm (scale, shift) from the per-tile partial sums convolutions wrote (conv2d(..., stats=True)); st2: the second
    part of a virtual channel concat."""
    Nn, C1, t1 = st1.shape[:3]
    C2, t2 = (st2.shape[1], st2.shape[2]) if st2 is not None else (0, 0)
    scale = torch.empty((Nn, C1 + C2), dtype=torch.float32, device=st1.device)
    shift = torch.empty_like(scale)
    N.check(N.lib().adm_groupnorm_finalize(N.ptr(st1), C1, t1, N.ptr(st2), C2, t2, Nn, hw, groups, float(eps), N.ptr(gamma),
                                           N.ptr(beta), N.ptr(scale), N.ptr(shift), N.stream_for(st1)))
    return scale, shift


def attention(qkv, head_dim):
    """qkv (N,3C,H,W) -> (N,C,H,W)."""
    _f32(qkv)
    Nn, C3, H, W = qkv.shape
    out = torch.empty((Nn, C3 // 3, H, W), dtype=torch.float32, device=qkv.device)
    N.check(N.lib().adm_attention(N.ptr(qkv), N.ptr(out), Nn, C3 // 3, H * W, head_dim, N.stream_for(qkv)))
    return out


# ---------------------------------------------------------------------------------------------- Transformer2DModel pieces
def layernorm_nct(x, gamma, beta, eps=1e-5):
    """LayerNorm over the channel axis of (N,C,H,W) for every token (BasicTransformerBlock.norm1/2/3)."""
    _f32(x)
    Nn, Cc = x.shape[:2]
    y = torch.empty_like(x)
    N.check(N.lib().adm_layernorm_nct(N.ptr(x), N.ptr(gamma), N.ptr(beta), N.ptr(y), Nn, Cc, x[0, 0].numel(), eps,
                                      N.stream_for(x)))
    return y


def geglu(x):
    """(N,2*C4,H,W) = [h | gate] -> (N,C4,H,W) = h * gelu(gate)."""
    _f32(x)
    Nn, C2 = x.shape[:2]
    out = torch.empty((Nn, C2 // 2) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    N.check(N.lib().adm_geglu(N.ptr(x), N.ptr(out), Nn, C2 // 2, x[0, 0].numel(), N.stream_for(x)))
    return out


def cross_attention(q, ctx, wk, wv, head_dim):
    """q (N,C,H,W), ctx (N,S,Dc), to_k/to_v weights (C,Dc) -> (N,C,H,W)."""
    _f32(q), _f32(ctx)
    Nn, Cc = q.shape[:2]
    out = torch.empty_like(q)
    N.check(N.lib().adm_cross_attention(N.ptr(q), N.ptr(ctx), N.ptr(wk), N.ptr(wv), N.ptr(out), Nn, Cc, q[0, 0].numel(),
                                        ctx.shape[1], ctx.shape[2], head_dim, N.stream_for(q)))
    return out


def layernorm_nct_backward(x, dy, gamma, eps=1e-5):
    """-> (dx, dgamma, dbeta) of layernorm_nct."""
    _f32(x), _f32(dy)
    Nn, Cc = x.shape[:2]
    T = x[0, 0].numel()
    dx = torch.empty_like(x)
    stats = torch.empty(2 * Nn * T, dtype=torch.float32, device=x.device)
    dg, db = torch.zeros_like(gamma), torch.zeros_like(gamma)
    N.check(N.lib().adm_layernorm_nct_backward(N.ptr(x), N.ptr(dy), N.ptr(gamma), N.ptr(dx), 0, N.ptr(stats), N.ptr(dg),
                                               N.ptr(db), Nn, Cc, T, eps, N.stream_for(x)))
    return dx, dg, db


def geglu_backward(x, dy):
    _f32(x), _f32(dy)
    Nn, C2 = x.shape[:2]
    dx = torch.empty_like(x)
    N.check(N.lib().adm_geglu_backward(N.ptr(x), N.ptr(dy), N.ptr(dx), Nn, C2 // 2, x[0, 0].numel(), N.stream_for(x)))
    return dx


def cross_attention_backward(q, ctx, wk, wv, dy, head_dim):
    """-> (dq, dWk, dWv) of cross_attention."""
    _f32(q), _f32(dy)
    Nn, Cc = q.shape[:2]
    dq = torch.empty_like(q)
    dwk, dwv = torch.zeros_like(wk), torch.zeros_like(wv)
    N.check(N.lib().adm_cross_attention_backward(N.ptr(q), N.ptr(ctx), N.ptr(wk), N.ptr(wv), N.ptr(dy), N.ptr(dq), N.ptr(dwk),
                                                 N.ptr(dwv), Nn, Cc, q[0, 0].numel(), ctx.shape[1], ctx.shape[2], head_dim,
                                                 N.stream_for(q)))
    return dq, dwk, dwv


def attention_backward_blocked(qkv, dout, head_dim, block=0):
    _f32(qkv), _f32(dout)
    Nn, C3, H, W = qkv.shape
    dqkv = torch.empty_like(qkv)
    stats = torch.empty(3 * Nn * (C3 // 3 // head_dim) * H * W, dtype=torch.float32, device=qkv.device)
    N.check(N.lib().adm_attention_backward_blocked(N.ptr(qkv), N.ptr(dout), N.ptr(dqkv), N.ptr(stats), Nn, C3 // 3, H * W,
                                                   head_dim, block, N.stream_for(qkv)))
    return dqkv


def attention_blocked(qkv, head_dim, key_block=0):
    """adm_attention with keys processed in blocks (online softmax): qkv (N,3C,H,W) -> (N,C,H,W)."""
    _f32(qkv)
    Nn, C3, H, W = qkv.shape
    out = torch.empty((Nn, C3 // 3, H, W), dtype=torch.float32, device=qkv.device)
    N.check(N.lib().adm_attention_blocked(N.ptr(qkv), N.ptr(out), Nn, C3 // 3, H * W, head_dim, key_block, N.stream_for(qkv)))
    return out


# ---------------------------------------------------------------------------------------------- backward ops (training)
def _conv_args(x1, wpacked, bias, ks, x2, up, stride, pad_lo, gn, act, Cout):
    Nn, C1, H, W = x1.shape
    a = N.ConvArgs()
    a.x1, a.C1 = N.ptr(x1), C1
    a.x2, a.C2 = (N.ptr(x2), x2.shape[1]) if x2 is not None else (None, 0)
    a.N, a.H, a.W = Nn, H, W
    a.up, a.stride, a.ks, a.pad_lo = int(up), stride, ks, pad_lo
    if gn is not None:
        a.gn_scale, a.gn_shift = N.ptr(gn[0]), N.ptr(gn[1])
    a.act = int(act)
    a.wpacked, a.bias, a.Cout = N.ptr(wpacked), N.ptr(bias), Cout
    return a


def sumpool2x2(x, out=None, accumulate=False):
    """(N,C,2H,2W) -> (N,C,H,W) sum over 2x2 blocks: backward of the folded nearest-x2 upsample."""
    _f32(x)
    Nn, Cc, H, W = x.shape
    out = torch.empty((Nn, Cc, H // 2, W // 2), dtype=torch.float32, device=x.device) if out is None else out
    N.check(N.lib().adm_sumpool2x2(N.ptr(x), N.ptr(out), H, W, Nn * Cc, int(accumulate), N.stream_for(x)))
    return out


def groupnorm_stats_ex(x1, gamma, beta, groups, eps, x2=None):
    """Like groupnorm_stats but also returns (N, groups, 2) [mean, rstd] for the backward pass."""
    Nn, C1 = x1.shape[:2]
    C2 = x2.shape[1] if x2 is not None else 0
    HW = x1[0, 0].numel()
    scale = torch.empty((Nn, C1 + C2), dtype=torch.float32, device=x1.device)
    shift = torch.empty_like(scale)
    mr = torch.empty((Nn, groups, 2), dtype=torch.float32, device=x1.device)
    N.check(N.lib().adm_groupnorm_stats_ex(N.ptr(x1), C1, N.ptr(x2), C2, Nn, HW, groups, float(eps), N.ptr(gamma),
                                           N.ptr(beta), N.ptr(scale), N.ptr(shift), N.ptr(mr), N.stream_for(x1)))
    return scale, shift, mr


def groupnorm_backward(x1, da, mean_rstd, gamma, beta, groups, act, x2=None):
    """Backward of a = act(GroupNorm(cat(x1,x2))): returns (dx1, dx2 or None, dgamma, dbeta)."""
    Nn, C1 = x1.shape[:2]
    C2 = x2.shape[1] if x2 is not None else 0
    HW = x1[0, 0].numel()
    dev = x1.device
    dg, db = torch.zeros(C1 + C2, device=dev), torch.zeros(C1 + C2, device=dev)
    s12 = torch.empty((Nn, groups, 2), dtype=torch.float32, device=dev)
    dx1 = torch.empty_like(x1)
    dx2 = torch.empty_like(x2) if x2 is not None else None
    N.check(N.lib().adm_groupnorm_backward(N.ptr(x1), C1, N.ptr(x2), C2, N.ptr(da.contiguous()), Nn, HW, groups,
                                           N.ptr(mean_rstd), N.ptr(gamma), N.ptr(beta), int(act), N.ptr(s12), N.ptr(dg),
                                           N.ptr(db), N.ptr(dx1), 0, N.ptr(dx2), 0, N.stream_for(x1)))
    return dx1, dx2, dg, db


def conv2d_wgrad(x1, dy, Cout, ks, x2=None, up=False, stride=1, pad_lo=1, gn=None, act=False):
    """dW (Cout,Cin,ks,ks) of the fused conv, with the load-path activation recomputed."""
    _f32(x1), _f32(dy)
    Ct = x1.shape[1] + (x2.shape[1] if x2 is not None else 0)
    a = _conv_args(x1, dy, None, ks, x2, up, stride, pad_lo, gn, act, Cout)  # wpacked unused by wgrad
    ws_n = N.lib().adm_conv_wgrad_workspace(C.byref(a))
    ws = torch.empty(ws_n, dtype=torch.float32, device=x1.device)
    dW = torch.zeros((Cout, Ct, ks, ks), dtype=torch.float32, device=x1.device)
    N.check(N.lib().adm_conv2d_wgrad(C.byref(a), N.ptr(dy), N.ptr(dW), 0, N.ptr(ws), N.stream_for(x1)))
    return dW


def blocked_image(x1, x2=None, gn=None, act=False, sums=False, zero_insert=False):
    """Blocked 16-bit operand image [n][C/8][H+2][W+2][8] (zero halo) of act(gn(concat(x1, x2))) — include/adm.h
    adm_blocked_apply.  Returns a (N, C/8, H+2, W+2, 8) bf16 tensor (binary16 bits under option conv_op16_f16);
    sums=True: also the per-(n, c) and per-c sums of the fp32 input."""
    _f32(x1)
    Nn, C1, H, W = x1.shape
    C2 = x2.shape[1] if x2 is not None else 0
    zi = 2 if zero_insert else 1        # zero_insert 1 / 2: the (2H, 2W) image with x(y, x) on pixel (2y + 1, 2x + 1) / (2y, 2x) (stride-2 backward)
    img = torch.zeros((Nn, (C1 + C2) // 8, zi * H + 2, zi * W + 2, 8), dtype=torch.bfloat16, device=x1.device)
    assert img.numel() * 2 == N.lib().adm_blocked_image_bytes(Nn, C1 + C2, zi * H, zi * W)
    nc = torch.zeros((Nn, C1 + C2), dtype=torch.float32, device=x1.device) if sums else None
    c = torch.zeros(C1 + C2, dtype=torch.float32, device=x1.device) if sums else None
    scr = torch.empty(N.lib().adm_blocked_sums_scratch(Nn, C1 + C2, H, W), dtype=torch.float32, device=x1.device) if sums else None
    N.check(N.lib().adm_blocked_apply(N.ptr(x1), C1, N.ptr(x2), C2, Nn, H, W, N.ptr(gn[0]) if gn is not None else None,
                                      N.ptr(gn[1]) if gn is not None else None, int(act), int(zero_insert), C.c_void_p(img.data_ptr()), N.ptr(scr),
                                      N.ptr(nc), C1 + C2, N.ptr(c), N.stream_for(x1)))
    return (img, nc, c) if sums else img


def conv2d_bf16_blocked(img, wb, Cout, bias=None, chan_add=None, residual=None, up=False, stats=False):
    """3x3 stride-1 convolution of a blocked image on 16-bit MFMA operands (adm_conv2d_bf16_blocked); wb from pack_bf16_weight
    (transposed=True with the image of dy: the data gradient); up: nearest x2 of the image folded in (Upsample2D.conv)."""
    Nn, Cg, Hp, Wp, _ = img.shape
    up = int(up)                        # 0 stride 1, 1 nearest x2 of the image folded in, 2 / 3 stride-2 output (pad (0,1,0,1) / padding 1)
    H, W = (Hp - 2) * (2 if up == 1 else 1), (Wp - 2) * (2 if up == 1 else 1)
    out = torch.empty((Nn, Cout, H // 2, W // 2) if up >= 2 else (Nn, Cout, H, W), dtype=torch.float32, device=img.device)
    ca, cas = (None, 0)
    if chan_add is not None:
        assert chan_add.dtype == torch.float32 and chan_add.stride(1) == 1
        ca, cas = C.c_void_p(chan_add.data_ptr()), chan_add.stride(0)
    st = torch.zeros((Nn, Cout, (H // 8) * (W // 32), 2), dtype=torch.float64, device=img.device) if stats else None
    assert not (stats and up >= 2)
    N.check(N.lib().adm_conv2d_bf16_blocked(C.c_void_p(img.data_ptr()), Cg * 8, Nn, H, W, C.c_void_p(wb.data_ptr()), Cout,
                                            N.ptr(bias), ca, cas, N.ptr(residual), N.ptr(out), int(up), N.ptr(st), N.stream_for(out)))
    return (out, st) if stats else out


def conv2d_wgrad_bf16_blocked(x_img, dy_img, up=False):
    """dW (Cout, Cin, 3, 3) from the blocked images of the activated input and of dy (adm_conv2d_wgrad_bf16_blocked);
    up: x_img is the half-resolution input of an Upsample2D convolution."""
    Nn, CgI = x_img.shape[:2]
    CgO, Hp, Wp = dy_img.shape[1:4]
    H, W, Cin, Cout = Hp - 2, Wp - 2, CgI * 8, CgO * 8
    ws_n = N.lib().adm_conv_wgrad_blocked_workspace(Cin, Cout, Nn, H, W)
    assert ws_n > 0, "shape not eligible"
    ws = torch.empty(ws_n, dtype=torch.float32, device=x_img.device)
    dW = torch.zeros((Cout, Cin, 3, 3), dtype=torch.float32, device=x_img.device)
    N.check(N.lib().adm_conv2d_wgrad_bf16_blocked(C.c_void_p(x_img.data_ptr()), Cin, C.c_void_p(dy_img.data_ptr()), Cout, Nn, H, W,
                                                  N.ptr(dW), 0, N.ptr(ws), int(up), N.stream_for(dW)))
    return dW


def chan_sums(dy):
    """(N,C,H,W) -> (per-(n,c) sums (N,C), per-c sums (C,))."""
    _f32(dy)
    Nn, Cc = dy.shape[:2]
    nc = torch.empty((Nn, Cc), dtype=torch.float32, device=dy.device)
    c = torch.zeros(Cc, dtype=torch.float32, device=dy.device)
    N.check(N.lib().adm_chan_sums(N.ptr(dy), Nn, Cc, dy[0, 0].numel(), N.ptr(nc), Cc, 0, N.ptr(c), N.stream_for(dy)))
    return nc, c


def attention_backward(qkv, dout, head_dim):
    _f32(qkv), _f32(dout)
    Nn, C3, H, W = qkv.shape
    dqkv = torch.empty_like(qkv)
    N.check(N.lib().adm_attention_backward(N.ptr(qkv), N.ptr(dout), N.ptr(dqkv), Nn, C3 // 3, H * W, head_dim,
                                           N.stream_for(qkv)))
    return dqkv


def linear_backward(dY, X, W, x_silu=False):
    """Y = b + W @ f(X) with f = silu if x_silu: returns (dW, db, dX)."""
    B, J = dY.shape
    K = X.shape[1]
    dW, db = torch.zeros((J, K), device=dY.device), torch.zeros(J, device=dY.device)
    dX = torch.empty((B, K), device=dY.device)
    N.check(N.lib().adm_linear_backward(N.ptr(dY.contiguous()), J, N.ptr(X), N.ptr(W), B, J, K, int(x_silu), N.ptr(dW),
                                        N.ptr(db), N.ptr(dX), N.stream_for(dY)))
    return dW, db, dX


def conv_small_cin_wgrad(x, dy):
    Nn, Ci, H, W = x.shape
    Co = dy.shape[1]
    dW = torch.zeros((Co, Ci, 3, 3), device=x.device)
    N.check(N.lib().adm_conv_small_cin_wgrad(N.ptr(x), Ci, Nn, H, W, N.ptr(dy), Co, N.ptr(dW), N.stream_for(x)))
    return dW


def conv_small_cout_backward(x, w, dy, gn=None, act=False):
    """conv_out class (Cout <= 4): returns (da, dW) with da the gradient w.r.t. the activated input."""
    Nn, Ci, H, W = x.shape
    Co = dy.shape[1]
    da = torch.empty_like(x)
    dW = torch.zeros((Co, Ci, 3, 3), device=x.device)
    N.check(N.lib().adm_conv_small_cout_backward(N.ptr(x), Ci, Nn, H, W, N.ptr(gn[0]) if gn else None,
                                                 N.ptr(gn[1]) if gn else None, int(act), N.ptr(w), N.ptr(dy), Co,
                                                 N.ptr(da), N.ptr(dW), N.stream_for(x)))
    return da, dW
