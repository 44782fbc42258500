"""Accuracy budget of the reduced-accuracy choices on the fp32 path, measured on the MI355X against float64 references:
  * per layer shape of the 256x256 UNet: GroupNorm + SiLU + 3x3 convolution (+ residual) through conv_wino6_kernel (Winograd F(4x4,3x3)),
    the F(2x2,3x3) kernels and the direct MFMA kernel — max|d| / max|ref| and rms(d) / rms(ref); the per-layer bar is 1e-4;
  * the UNet's attention (64 heads x d = 8, T = 256) and the flash self-attention of the conditional model (d = 64, T = 4096);
  * with ADM_LIB=tools/libadm_precise.so (a -DADM_PRECISE_MATH build: libm expf and IEEE division where the product uses v_exp_f32 / v_rcp_f32)
    the same numbers once more, plus the outputs saved for the whole-network comparison (UNet forward at B = 1, VAE decode).
Usage: python tools/accuracy_probe.py <tag>;  python tools/accuracy_probe.py compare <tagA> <tagB>"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
OUT = os.path.join(ROOT, "gpurun_out", "acc")
os.makedirs(OUT, exist_ok=True)

if len(sys.argv) > 1 and sys.argv[1] == "compare":
    a, b = (torch.load(os.path.join(OUT, t + ".pt")) for t in sys.argv[2:4])
    for k in a:
        d = (a[k].double() - b[k].double()).abs().max().item()
        print(f"{k}: max|{sys.argv[2]} - {sys.argv[3]}| = {d:.3e}  ({d / a[k].double().abs().max().item():.3e} of max|out| {a[k].abs().max().item():.3f})")
    sys.exit(0)

from audiodiffusion import _native, ops  # noqa: E402
from audiodiffusion.unet import UNet2DModel  # noqa: E402
from bench import CFG256  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "default"
_native.load(os.environ.get("ADM_LIB") or None)
lib = _native.lib()
dev = torch.device("cuda:0")
torch.set_num_threads(min(32, os.cpu_count() or 8))
SHAPES = [  # (C1, C2, H, Cout, up, residual): the 3x3 stride-1 layers of the 256x256 model the F(4x4) kernel can tile, and two it is kept from by default
    (128, 0, 256, 128, 0, 1), (128, 128, 256, 128, 0, 1), (256, 128, 128, 128, 0, 0), (128, 0, 128, 128, 1, 0),
    (256, 0, 64, 256, 0, 1), (256, 256, 64, 256, 0, 0), (512, 512, 16, 512, 0, 1),
]


def err(out, ref):
    d = out.double().cpu() - ref
    return d.abs().max().item() / ref.abs().max().item(), (d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()


print(f"== {tag}: library {os.environ.get('ADM_LIB', 'product')}")
print("| layer (C1+C2 -> Cout @HxW, up, residual) | F(4x4) max / rms | F(2x2) max / rms | direct MFMA max / rms |")
print("|---|---|---|---|")
for (C1, C2, H, Co, up, has_res) in SHAPES:
    g = torch.Generator().manual_seed(C1 + 7 * C2 + H + 3 * up)
    C = C1 + C2
    x = torch.randn(1, C, H, H, generator=g) * 1.5 + 0.3
    w = torch.randn(Co, C, 3, 3, generator=g) * (C * 9) ** -0.5
    b = torch.randn(Co, generator=g) * 0.1
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    Ho = 2 * H if up else H
    res = torch.randn(1, Co, Ho, Ho, generator=g) if has_res else None
    xd = x.double()
    h = torch.nn.functional.silu(torch.nn.functional.group_norm(xd, 32, gamma.double(), beta.double(), 1e-5))
    if up:
        h = torch.nn.functional.interpolate(h, scale_factor=2.0, mode="nearest")
    ref = torch.nn.functional.conv2d(h, w.double(), b.double(), padding=1)
    if has_res:
        ref = ref + res.double()
    x1 = x[:, :C1].contiguous().to(dev)
    x2 = x[:, C1:].contiguous().to(dev) if C2 else None
    wd = w.to(dev)
    _native.check(lib.adm_set_option(b"conv_wino", 4))     # (the Winograd image's layout follows the mode in force when it is packed)
    wp, wu = ops.pack_conv_weight(wd), ops.pack_winograd_weight(wd)
    gn = ops.groupnorm_stats(x1, gamma.to(dev), beta.to(dev), 32, 1e-5, x2=x2)
    cells = []
    for opts in (((b"conv_wino", 4), (b"wino6", 2)), ((b"conv_wino", 4), (b"wino6", 0)), ((b"conv_wino", 0), (b"wino6", 0))):
        for k, v in opts:
            _native.check(lib.adm_set_option(k, v))
        out = ops.conv2d(x1, wp, b.to(dev), 3, x2=x2, up=bool(up), gn=gn, act=True, wino=wu, residual=None if res is None else res.to(dev))
        var = lib.adm_last_conv_variant()
        m, r = err(out, ref)
        cells.append(f"{m:.2e} / {r:.2e} ({var})")
    print(f"| {C1}+{C2} -> {Co} @{H}x{H}, up {up}, res {has_res} | " + " | ".join(cells) + " |", flush=True)
_native.check(lib.adm_set_option(b"conv_wino", -1))
_native.check(lib.adm_set_option(b"wino6", -1))

print("| attention | max|d| / max|ref| | rms |")
for (Nn, Cc, T, d) in ((2, 512, 256, 8), (1, 320, 4096, 64), (1, 512, 1024, 64)):
    g = torch.Generator().manual_seed(T + d)
    qkv = torch.randn(Nn, 3 * Cc, T, 1, generator=g) * 1.2
    q, k, v = (t.double().view(Nn, Cc // d, d, T) for t in qkv[..., 0].chunk(3, dim=1))
    p = torch.softmax(torch.einsum("nhdq,nhdk->nhqk", q, k) * d ** -0.5, dim=-1)
    ref = torch.einsum("nhqk,nhdk->nhdq", p, v).reshape(Nn, Cc, T, 1)
    out = ops.attention(qkv.to(dev), d)
    m, r = err(out, ref)
    print(f"| {Cc // d} heads x d = {d}, T = {T} | {m:.2e} | {r:.2e} |", flush=True)

# whole networks, for the comparison of two libraries (the oracle comparison of each is in tests/test_full_size.py)
save = {}
unet = UNet2DModel(**CFG256).init_random(0).to(dev) if hasattr(UNet2DModel, "to") else UNet2DModel(**CFG256).init_random(0)
x = torch.randn(1, 1, 256, 256, generator=torch.Generator().manual_seed(1)).to(dev)
save["UNet2DModel 256x256 forward, B = 1, t = 500 (random init)"] = unet(x, 500)["sample"].float().cpu()
try:
    from audiodiffusion.vae import AutoencoderKL
    vae = AutoencoderKL().init_random(0)
    z = torch.randn(1, vae.config.latent_channels, 32, 32, generator=torch.Generator().manual_seed(2)).to(dev)
    save["AutoencoderKL decode of a 32x32 latent (mid-block attention T = 1024)"] = vae.decode(z)["sample"].float().cpu()
except Exception as e:  # noqa: BLE001
    print("VAE leg skipped:", type(e).__name__, e)
torch.save(save, os.path.join(OUT, tag + ".pt"))
print("saved", list(save))
