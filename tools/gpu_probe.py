"""GPU probe (run on the MI355X box via gpurun): production-shape conv timings, whole-UNet forward timing,
full-size parity vs the oracle. Writes gpurun_out/probe.log."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
from audiodiffusion import _native, ops  # noqa: E402
from audiodiffusion.unet import UNet2DModel  # noqa: E402

_native.load()
dev = torch.device("cuda:0")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(ROOT, "gpurun_out", "probe.log"), "a")


def log(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    LOG.write(s + "\n")
    LOG.flush()


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


CFG256 = dict(sample_size=256, in_channels=1, out_channels=1, layers_per_block=2,
              block_out_channels=(128, 128, 256, 256, 512, 512),
              down_block_types=("DownBlock2D",) * 4 + ("AttnDownBlock2D", "DownBlock2D"),
              up_block_types=("UpBlock2D", "AttnUpBlock2D") + ("UpBlock2D",) * 4)


def conv_shapes():
    B = int(os.environ.get("PROBE_B", "16"))
    shapes = [  # (C1, C2, H, W, Cout, ks, stride, up)
        (128, 0, 256, 256, 128, 3, 1, 0), (128, 128, 256, 256, 128, 3, 1, 0), (128, 0, 128, 128, 128, 3, 1, 0),
        (128, 0, 64, 64, 256, 3, 1, 0), (256, 0, 64, 64, 256, 3, 1, 0), (256, 0, 32, 32, 256, 3, 1, 0),
        (256, 0, 16, 16, 512, 3, 1, 0), (512, 0, 16, 16, 512, 3, 1, 0), (512, 0, 8, 8, 512, 3, 1, 0),
        (512, 512, 8, 8, 512, 3, 1, 0), (512, 512, 16, 16, 512, 3, 1, 0), (256, 256, 32, 32, 256, 3, 1, 0),
        (128, 0, 256, 256, 128, 3, 2, 0), (128, 0, 128, 128, 128, 3, 1, 1), (256, 128, 128, 128, 128, 1, 1, 0),
        (512, 0, 16, 16, 1536, 1, 1, 0),
    ]
    if os.environ.get("PROBE_SHORTCUTS"):     # the conv_shortcut class: 1x1, no GroupNorm / activation on the load path
        shapes = [(128, 128, 256, 256, 128, 1, 1, -1), (256, 128, 128, 128, 128, 1, 1, -1), (256, 256, 64, 64, 256, 1, 1, -1),
                  (512, 512, 16, 16, 512, 1, 1, -1)]
    for (C1, C2, H, W, Co, ks, st, up) in shapes:
        x1 = torch.randn(B, C1, H, W, device=dev)
        x2 = torch.randn(B, C2, H, W, device=dev) if C2 else None
        w = torch.randn(Co, C1 + C2, ks, ks, device=dev) * 0.02
        wp = ops.pack_conv_weight(w)
        b = torch.randn(Co, device=dev)
        gamma, beta = torch.ones(C1 + C2, device=dev), torch.zeros(C1 + C2, device=dev)
        gn = ops.groupnorm_stats(x1, gamma, beta, 32, 1e-5, x2=x2)
        wu = ops.pack_winograd_weight(w) if (ks == 3 and st == 1 and os.environ.get('ADM_CONV_WINO', '3') in ('1', '2', '3')) else None
        if up < 0:
            f = lambda: ops.conv2d(x1, wp, b, ks, x2=x2)  # noqa: E731
        else:
            f = lambda: ops.conv2d(x1, wp, b, ks, x2=x2, up=bool(up), stride=st, gn=gn, act=True, wino=wu)  # noqa: E731
        out = f()
        dt = timeit(f, iters=5, warm=2)
        flops = 2.0 * out.numel() * (C1 + C2) * ks * ks
        g = lambda: ops.groupnorm_stats(x1, gamma, beta, 32, 1e-5, x2=x2)  # noqa: E731
        dg = timeit(g, iters=5, warm=2)
        gbytes = 4.0 * (x1.numel() + (x2.numel() if x2 is not None else 0))
        log(f"conv B={B} {C1}+{C2}@{H}x{W}->{Co} k{ks} s{st} up{up}: {dt*1e3:8.3f} ms {flops/dt/1e12:7.2f} TF/s |"
            f" gn_stats {dg*1e3:7.3f} ms {gbytes/dg/1e12:5.2f} TB/s")
        if os.environ.get("PROBE_CHECK", "1") == "1" and H <= 64 and up >= 0:
            import torch.nn.functional as F
            xc = torch.cat([x1, x2], 1) if x2 is not None else x1
            xr = F.silu(F.group_norm(xc, 32, gamma, beta, 1e-5))
            if up:
                xr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
            ref = F.conv2d(xr, w, b, stride=st, padding=ks // 2)
            log("   vs torch(MIOpen) relerr", float((out - ref).abs().max() / ref.abs().max()))
        del x1, x2, w, wp, out


def unet_probe():
    m = UNet2DModel(**CFG256).init_random(0)
    for B in (1, 4, 16, 32):
        x = torch.randn(B, 1, 256, 256, device=dev)
        t = torch.tensor(500)
        f = lambda: m(x, t)  # noqa: E731
        f()
        torch.cuda.synchronize()
        dt = timeit(f, iters=3, warm=1)
        log(f"unet fwd B={B}: {dt*1e3:9.2f} ms  {0.496*B/dt:7.2f} TF/s  ws={_native.lib().adm_unet_workspace_bytes(m._handle)/2**30:.2f} GiB")
    return m


def parity(m):
    from oracle.unet import UNet2DModel as OU
    ref = OU(**CFG256).eval()
    ref.load_state_dict(m.state_dict())
    g = torch.Generator().manual_seed(42)
    x = torch.randn(1, 1, 256, 256, generator=g)
    for t in (980, 20):
        t0 = time.perf_counter()
        with torch.no_grad():
            r = ref(x, torch.tensor(t))["sample"]
        tc = time.perf_counter() - t0
        o = m(x.to(dev), torch.tensor(t))["sample"].cpu()
        log(f"full-size parity t={t}: max|d|={float((o - r).abs().max()):.3e} max|ref|={float(r.abs().max()):.3f} cpu_oracle={tc:.2f}s")


if __name__ == "__main__":
    what = sys.argv[1:] or ["conv", "unet", "parity"]
    log("device", torch.cuda.get_device_name(0), "probe", what)
    if "conv" in what:
        conv_shapes()
    m = None
    if "unet" in what or "parity" in what:
        m = unet_probe()
    if "parity" in what:
        parity(m)


def vae_probe():
    """Config-4 class shapes: AutoencoderKL (128,256,512,512) at 256x256 -> latent 32x32; parity vs oracle at B=1."""
    from audiodiffusion.vae import AutoencoderKL
    from oracle.vae import AutoencoderKL as OV
    cfg = dict(sample_size=(256, 256), in_channels=1, out_channels=1, latent_channels=1, layers_per_block=2,
               block_out_channels=(128, 256, 512, 512), down_block_types=("DownEncoderBlock2D",) * 4,
               up_block_types=("UpDecoderBlock2D",) * 4)
    v = AutoencoderKL(**cfg).init_random(0)
    ref = OV(**cfg).eval()
    ref.load_state_dict(v.state_dict())
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 1, 256, 256, generator=g)
    with torch.no_grad():
        d = ref.encode(x).latent_dist
        nz = torch.randn(d.mean.shape, generator=g)
        rz = d.sample(noise=nz)
        rd = ref.decode(rz)["sample"]
    mz = v.encode(x.to(dev)).latent_dist.sample(noise=nz.to(dev))
    md = v.decode(rz.to(dev))["sample"]
    log(f"vae parity: enc max|d|={float((mz.cpu()-rz).abs().max()):.3e} (max|z|={float(rz.abs().max()):.3f}) "
        f"dec max|d|={float((md.cpu()-rd).abs().max()):.3e} (max|ref|={float(rd.abs().max()):.3f})")
    for B in (1, 16):
        xb = torch.randn(B, 1, 256, 256, device=dev)
        zb = torch.randn(B, 1, 32, 32, device=dev)
        te = timeit(lambda: v.encode(xb).latent_dist.mode(), iters=3, warm=1)
        td = timeit(lambda: v.decode(zb), iters=3, warm=1)
        log(f"vae B={B}: encode {te*1e3:8.2f} ms ({0.272*B/te:6.1f} TF/s)  decode {td*1e3:8.2f} ms ({0.622*B/td:6.1f} TF/s)")


if __name__ == "__main__" and "vae" in sys.argv[1:]:
    vae_probe()


def mel_probe():
    import numpy as np
    from audiodiffusion.mel import Mel
    from oracle import mel as omel
    m, om = Mel(), omel.Mel()
    rng = np.random.default_rng(0)
    for B in (1, 32, 256):
        ys = (0.3 * rng.standard_normal((B, m.slice_size))).astype(np.float32)
        t = torch.from_numpy(ys).to(dev)
        out = torch.empty((B, 256, 256), dtype=torch.uint8, device=dev)
        h = m._ensure_handle()
        from audiodiffusion import _native as N
        f = lambda: N.check(N.lib().adm_mel_forward(h, N.ptr(t), 0, B, ys.shape[1], ys.shape[1], N.ptr(out), None))  # noqa: E731
        tf = timeit(f, iters=5, warm=2)
        imgs = out.cpu().numpy()
        img_t = torch.from_numpy(imgs).to(dev)
        phase = torch.rand((B, 1025, 256), dtype=torch.float64, device=dev)
        aud = torch.empty((B, 130560), dtype=torch.float32, device=dev)
        g = lambda: N.check(N.lib().adm_mel_inverse(h, N.ptr(img_t), N.ptr(phase), B, 256, N.ptr(aud), None, None, None))  # noqa: E731
        ti = timeit(g, iters=2, warm=1)
        log(f"mel B={B}: audio->image {tf*1e3:8.3f} ms ({B/tf:9.1f} slices/s, {B*131071*4/tf/1e9:6.1f} GB/s in)  "
            f"image->audio {ti*1e3:9.2f} ms ({B/ti:8.1f} clips/s)")
    # CPU oracle, single call each (the reference runs these serially per image)
    y = (0.3 * rng.standard_normal(m.slice_size)).astype(np.float32)
    om.load_audio(raw_audio=y)
    t0 = time.perf_counter(); img = om.audio_slice_to_image(0); t1 = time.perf_counter()
    om.image_to_audio(img); t2 = time.perf_counter()
    log(f"mel CPU oracle: audio->image {1e3*(t1-t0):.1f} ms, image->audio {1e3*(t2-t1):.1f} ms per call")


if __name__ == "__main__" and "mel" in sys.argv[1:]:
    mel_probe()


def train_probe():
    """Optimizer side at the real parameter count (113.67 M): bytes/s of the fused AdamW+EMA and the norm pass."""
    from audiodiffusion import training as T
    n = 113_668_609
    p = torch.randn(n, device=dev)
    g = torch.randn(n, device=dev) * 0.01
    opt = T.AdamW(p)
    ema = T.EMAModel(p)
    f = lambda: opt.step(g, clip=T.clip_grad_norm_(g, 1.0), ema=ema, ema_decay=0.999)  # noqa: E731
    dt = timeit(f, iters=10, warm=2)
    log(f"train: clip-norm + AdamW + EMA over {n/1e6:.1f}M params: {dt*1e3:.3f} ms  ({(4+36)*n/dt/1e12:.2f} TB/s algorithmic)")


if __name__ == "__main__" and "train" in sys.argv[1:]:
    train_probe()


def trainstep_probe():
    """Full-size training step (config 5 shape; PROBE_MP=bf16 for mixed precision): forward+backward+clip+AdamW+EMA, B per GPU from PROBE_B."""
    from audiodiffusion import training as T
    B = int(os.environ.get("PROBE_B", "8"))
    m = UNet2DModel(**CFG256).init_random(0)
    flat, grads = m.enable_training(mixed_precision=os.environ.get("PROBE_MP", "no"))
    opt, ema = T.AdamW(flat), T.EMAModel(flat)
    x = torch.randn(B, 1, 256, 256, device=dev)
    tgt = torch.randn(B, 1, 256, 256, device=dev)
    ts = torch.randint(0, 1000, (B,))

    def step():
        loss = m.train_step(x, ts, tgt)
        clip = T.clip_grad_norm_(grads, 1.0)
        opt.step(grads, clip=clip, ema=ema, ema_decay=ema.next_decay())
        m.refresh_weights()
        return loss

    l0 = float(step())
    torch.cuda.synchronize()
    dt = timeit(step, iters=3, warm=1)
    l1 = float(step())
    log(f"train step B={B}: {dt*1e3:9.2f} ms  {B/dt:7.2f} samples/s  {3*0.496*B/dt:7.2f} TF/s(3x fwd flops)  loss {l0:.4f} -> {l1:.4f}  "
        f"mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB torch + native arena")
    if os.environ.get("PROBE_CHECK", "1") == "1" and B <= 2:
        import torch.nn.functional as F
        from oracle.unet import UNet2DModel as OU
        m2 = UNet2DModel(**CFG256).init_random(1)
        ref = OU(**CFG256)
        ref.load_state_dict(m2.state_dict())
        flat2, grads2 = m2.enable_training()
        xc, tc = x[:1].cpu(), tgt[:1].cpu()
        lr = F.mse_loss(ref(xc, ts[:1])["sample"], tc)
        lr.backward()
        lm = m2.train_step(x[:1].contiguous(), ts[:1], tgt[:1].contiguous())
        gmax = max(float(p.grad.abs().max()) for p in ref.parameters())
        worst = 0.0
        for name, p in ref.named_parameters():
            off = m2.flat.offsets[name][0]
            got = grads2[off:off + p.numel()].view(p.shape).cpu()
            worst = max(worst, float((got - p.grad).abs().max()) / max(float(p.grad.abs().max()), 1e-3 * gmax))
        log(f"full-size grad parity B=1: loss {float(lm):.6f} vs {float(lr.detach()):.6f}; worst per-tensor rel err {worst:.2e}")


if __name__ == "__main__" and "trainstep" in sys.argv[1:]:
    trainstep_probe()
