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


def pack_winograd_weight(w):
    """(Cout,Cin,3,3) -> Winograd-domain U = G g G^T as [Cin][16][Cout]."""
    _f32(w)
    co, ci = w.shape[:2]
    wu = torch.empty((ci, 16, co), dtype=torch.float32, device=w.device)
    N.check(N.lib().adm_pack_winograd_weight(N.ptr(w), N.ptr(wu), co, ci, N.stream_for(w)))
    return wu


def conv2d(x1, wpacked, bias, ks, x2=None, up=False, stride=1, pad_lo=1, gn=None, act=False, chan_add=None,
           residual=None, wino=None):
    """Fused convolution (see include/adm.h adm_conv_args)."""
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
    a.up, a.stride, a.ks, a.pad_lo = int(up), stride, ks, pad_lo
    if gn is not None:
        a.gn_scale, a.gn_shift = N.ptr(gn[0]), N.ptr(gn[1])
    a.act = int(act)
    a.wpacked, a.bias, a.Cout = N.ptr(wpacked), N.ptr(bias), Cout
    if chan_add is not None:
        assert chan_add.dtype == torch.float32 and chan_add.stride(1) == 1
        a.chan_add, a.chan_add_stride = C.c_void_p(chan_add.data_ptr()), chan_add.stride(0)
    a.residual = N.ptr(residual)
    a.wino_packed = N.ptr(wino)
    a.out = N.ptr(out)
    N.check(N.lib().adm_conv2d(C.byref(a), N.stream_for(x1)))
    return out


def attention(qkv, head_dim):
    """qkv (N,3C,H,W) -> (N,C,H,W)."""
    _f32(qkv)
    Nn, C3, H, W = qkv.shape
    out = torch.empty((Nn, C3 // 3, H, W), dtype=torch.float32, device=qkv.device)
    N.check(N.lib().adm_attention(N.ptr(qkv), N.ptr(out), Nn, C3 // 3, H * W, head_dim, N.stream_for(qkv)))
    return out
