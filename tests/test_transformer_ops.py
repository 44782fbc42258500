"""Transformer2DModel pieces of UNet2DConditionModel (scripts/train_unet.py:139-159) vs plain torch fp32."""
import pytest
import torch
import torch.nn.functional as F

from native_backend import BACKENDS, select
from test_kernels import _rand, _relerr


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("shape", [(2, 32, 4, 8), (1, 96, 16, 16), (3, 64, 1, 5), (2, 128, 8, 16), (1, 320, 16, 16)])
def test_layernorm_over_channels(backend, shape):
    dev = select(backend)
    from audiodiffusion import ops
    x = _rand(shape, 1, dev) * 2 + 0.3
    g, b = _rand((shape[1],), 2, dev) + 1, _rand((shape[1],), 3, dev)
    y = ops.layernorm_nct(x, g, b)
    ref = F.layer_norm(x.cpu().permute(0, 2, 3, 1), (shape[1],), g.cpu(), b.cpu(), 1e-5).permute(0, 3, 1, 2)
    assert _relerr(y, ref) < 2e-6


@pytest.mark.parametrize("backend", BACKENDS)
def test_geglu(backend):
    dev = select(backend)
    from audiodiffusion import ops
    x = _rand((2, 64, 4, 8), 1, dev) * 3
    h, gate = x.cpu().chunk(2, dim=1)
    assert _relerr(ops.geglu(x), h * F.gelu(gate)) < 2e-6


def _mha(q, k, v, heads):
    """q (N,C,Tq), k/v (N,C,Tk) channel-major -> (N,C,Tq): heads of C/heads consecutive channels."""
    Nn, Cc, Tq = q.shape
    d = Cc // heads

    def sp(t):
        return t.view(Nn, heads, d, -1).transpose(2, 3)             # (N, heads, T, d)

    p = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * d ** -0.5, dim=-1)
    return (p @ sp(v)).transpose(2, 3).reshape(Nn, Cc, Tq)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("C,heads,S", [(32, 8, 1), (64, 8, 3), (128, 8, 5), (32, 2, 2), (64, 2, 2), (128, 2, 3)])
def test_cross_attention_on_encoding(backend, C, heads, S):
    dev = select(backend)
    from audiodiffusion import ops
    Nn, H, W, Dc = 2, 4, 8, 12
    q = _rand((Nn, C, H, W), 1, dev)
    ctx = _rand((Nn, S, Dc), 2, dev)
    wk, wv = _rand((C, Dc), 3, dev, scale=Dc ** -0.5), _rand((C, Dc), 4, dev, scale=Dc ** -0.5)
    out = ops.cross_attention(q, ctx, wk, wv, C // heads)
    k = (ctx.cpu() @ wk.cpu().T).transpose(1, 2)                     # (N, C, S)
    v = (ctx.cpu() @ wv.cpu().T).transpose(1, 2)
    ref = _mha(q.cpu().reshape(Nn, C, H * W), k, v, heads).reshape(Nn, C, H, W)
    assert _relerr(out, ref) < 5e-6
    if S == 1:       # one key: the softmax is 1 and every token receives V (what the reference's seq_length-1 encoding does)
        assert _relerr(out, v.reshape(Nn, C, 1, 1).expand(Nn, C, H, W)) < 2e-6


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("C,heads,HW,key_block", [(32, 2, (8, 8), 16), (64, 8, (4, 24), 32), (32, 8, (8, 8), 0),
                                                   (64, 2, (8, 8), 24), (128, 2, (4, 8), 16)])
def test_self_attention_key_blocks_match_one_pass(backend, C, heads, HW, key_block):
    """Online softmax over key blocks (needed at 64x64 latents: 4096 tokens) == the one-pass kernel == torch."""
    dev = select(backend)
    from audiodiffusion import ops
    Nn = 2
    qkv = _rand((Nn, 3 * C) + HW, 1, dev) * 1.5
    out = ops.attention_blocked(qkv, C // heads, key_block)
    T = HW[0] * HW[1]
    q, k, v = qkv.cpu().reshape(Nn, 3, C, T).unbind(1)
    ref = _mha(q, k, v, heads).reshape(Nn, C, *HW)
    assert _relerr(out, ref) < 5e-6
    assert _relerr(out, ops.attention(qkv, C // heads)) < 5e-6


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("C,heads,HW", [(128, 8, (16, 16)), (256, 8, (16, 16)), (512, 8, (8, 16)), (128, 8, (16, 32)), (64, 4, (16, 16))],
                         ids=["d16-T256", "d32-T256", "d64-T128", "d16-T512", "d16-4heads"])
def test_self_attention_on_the_matrix_pipe(backend, C, heads, HW):
    """Round 5: for head dimensions 16 / 32 / 64 `adm_attention` runs the flash form on v_mfma_f32_16x16x4_f32 (the conditional UNet's
    Transformer2DModel blocks: 8 heads at 128 / 256 / 512 channels; 4096 tokens at the 512-resolution latent size). Against torch's
    softmax attention and against the VALU online-softmax kernel (another program, another summation order) on the same tensor."""
    dev = select(backend)
    from audiodiffusion import _native, ops
    Nn = 2
    T = HW[0] * HW[1]
    assert _native.lib().adm_attention_mfma_eligible(C, T, C // heads) == 1
    qkv = _rand((Nn, 3 * C) + HW, 3, dev) * 1.5
    out = ops.attention(qkv, C // heads)
    q, k, v = qkv.cpu().reshape(Nn, 3, C, T).unbind(1)
    ref = _mha(q, k, v, heads).reshape(Nn, C, *HW)
    assert _relerr(out, ref) < 5e-6, _relerr(out, ref)
    assert _relerr(out, ops.attention_blocked(qkv, C // heads, 64)) < 5e-6
    # the shapes the kernel does not tile stay on the VALU kernels
    assert _native.lib().adm_attention_mfma_eligible(64, 256, 8) == 0 and _native.lib().adm_attention_mfma_eligible(128, 192, 16) == 0


# ---------------------------------------------------------------- backward passes vs torch autograd
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("shape", [(2, 32, 4, 8), (1, 96, 16, 16)])
def test_layernorm_backward(backend, shape):
    dev = select(backend)
    from audiodiffusion import ops
    x = (_rand(shape, 1, "cpu") * 2 + 0.3).requires_grad_(True)
    g, b = (_rand((shape[1],), 2, "cpu") + 1).requires_grad_(True), _rand((shape[1],), 3, "cpu").requires_grad_(True)
    y = F.layer_norm(x.permute(0, 2, 3, 1), (shape[1],), g, b, 1e-5).permute(0, 3, 1, 2)
    dy = _rand(shape, 4, "cpu")
    y.backward(dy)
    dx, dg, db = ops.layernorm_nct_backward(x.detach().to(dev), dy.to(dev), g.detach().to(dev))
    assert _relerr(dx, x.grad) < 1e-5 and _relerr(dg, g.grad) < 1e-5 and _relerr(db, b.grad) < 1e-5


@pytest.mark.parametrize("backend", BACKENDS)
def test_geglu_backward(backend):
    dev = select(backend)
    from audiodiffusion import ops
    x = (_rand((2, 64, 4, 8), 1, "cpu") * 3).requires_grad_(True)
    h, gate = x.chunk(2, dim=1)
    dy = _rand((2, 32, 4, 8), 2, "cpu")
    (h * F.gelu(gate)).backward(dy)
    assert _relerr(ops.geglu_backward(x.detach().to(dev), dy.to(dev)), x.grad) < 2e-6


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("C,heads,S", [(32, 8, 1), (64, 8, 3), (32, 2, 2), (64, 2, 2), (128, 2, 3)])
def test_cross_attention_backward(backend, C, heads, S):
    dev = select(backend)
    from audiodiffusion import ops
    Nn, H, W, Dc = 2, 4, 8, 12
    q = _rand((Nn, C, H, W), 1, "cpu").requires_grad_(True)
    ctx = _rand((Nn, S, Dc), 2, "cpu")
    wk = _rand((C, Dc), 3, "cpu", scale=Dc ** -0.5).requires_grad_(True)
    wv = _rand((C, Dc), 4, "cpu", scale=Dc ** -0.5).requires_grad_(True)
    out = _mha(q.reshape(Nn, C, H * W), (ctx @ wk.T).transpose(1, 2), (ctx @ wv.T).transpose(1, 2), heads).reshape(Nn, C, H, W)
    dy = _rand((Nn, C, H, W), 5, "cpu")
    out.backward(dy)
    dq, dwk, dwv = ops.cross_attention_backward(q.detach().to(dev), ctx.to(dev), wk.detach().to(dev), wv.detach().to(dev),
                                                dy.to(dev), C // heads)
    assert _relerr(dwv, wv.grad) < 1e-5
    if S == 1:       # a softmax over one key has no gradient: dq = 0, dWk = 0
        assert float(dq.abs().max()) == 0 and float(dwk.abs().max()) < 1e-6 * float(wv.grad.abs().max())
    else:
        assert _relerr(dq, q.grad) < 1e-5 and _relerr(dwk, wk.grad) < 2e-5


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("C,heads,HW,block", [(32, 2, (8, 8), 16), (64, 8, (4, 24), 32), (64, 4, (8, 8), 0),
                                               (64, 2, (8, 8), 24), (128, 2, (4, 8), 16)])
def test_self_attention_backward_in_blocks(backend, C, heads, HW, block):
    dev = select(backend)
    from audiodiffusion import ops
    Nn, T = 2, HW[0] * HW[1]
    qkv = (_rand((Nn, 3 * C) + HW, 1, "cpu") * 1.5).requires_grad_(True)
    q, k, v = qkv.reshape(Nn, 3, C, T).unbind(1)
    dout = _rand((Nn, C) + HW, 2, "cpu")
    _mha(q, k, v, heads).reshape(Nn, C, *HW).backward(dout)
    got = ops.attention_backward_blocked(qkv.detach().to(dev), dout.to(dev), C // heads, block)
    assert _relerr(got, qkv.grad) < 2e-5
    if C // heads <= 32:
        assert _relerr(got, ops.attention_backward(qkv.detach().to(dev), dout.to(dev), C // heads)) < 2e-5
