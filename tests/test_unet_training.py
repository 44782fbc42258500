"""Whole-UNet training step parity (SURVEY.md §7.1 phase 10 exit test): loss and d loss/d parameter of the native
forward+backward vs torch autograd on the oracle UNet, identical weights / inputs; then one fused AdamW+EMA step."""
import pytest
import torch
import torch.nn.functional as F

from native_backend import BACKENDS, select
from oracle.unet import UNet2DModel as OracleUNet

TINY = dict(sample_size=16, in_channels=1, out_channels=1, layers_per_block=1, block_out_channels=(32, 64),
            down_block_types=("DownBlock2D", "AttnDownBlock2D"), up_block_types=("AttnUpBlock2D", "UpBlock2D"))
TINY3 = dict(sample_size=(8, 16), in_channels=1, out_channels=1, layers_per_block=2, block_out_channels=(32, 32, 64),
             down_block_types=("DownBlock2D", "AttnDownBlock2D", "DownBlock2D"),
             up_block_types=("UpBlock2D", "AttnUpBlock2D", "UpBlock2D"))


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("cfg,B", [(TINY, 2), (TINY3, 3)], ids=["tiny2", "tiny3"])
def test_forward_backward_matches_autograd(backend, cfg, B):
    dev = select(backend)
    from audiodiffusion.unet import UNet2DModel
    torch.manual_seed(0)
    ref = OracleUNet(**cfg)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "norm" in n:
                p.add_(0.2 * torch.randn_like(p))
    mine = UNet2DModel(**cfg).load_state_dict(ref.state_dict())
    flat, grads = mine.enable_training()
    ss = cfg["sample_size"]
    hw = (ss, ss) if isinstance(ss, int) else ss
    g = torch.Generator().manual_seed(1)
    x = torch.randn((B, 1) + tuple(hw), generator=g)
    tgt = torch.randn((B, 1) + tuple(hw), generator=g)
    ts = torch.tensor([5, 500, 999][:B])
    loss_ref = F.mse_loss(ref(x, ts)["sample"], tgt)
    loss_ref.backward()
    loss = mine.train_step(x.to(dev), ts, tgt.to(dev))
    assert abs(float(loss) - float(loss_ref.detach())) <= 1e-5 * float(loss_ref.detach())
    # Tolerance 2e-4 relative to each tensor's own gradient scale; gradients that are mathematically zero (a bias feeding
    # a GroupNorm whose groups are single channels, to_k.bias under the softmax) are 1e-9 noise on both sides and are
    # compared against the global gradient scale instead.
    gmax = max(float(p.grad.abs().max()) for p in ref.parameters())
    checked = 0
    for name, p in ref.named_parameters():
        off = mine.flat.offsets[name][0]
        got = grads[off:off + p.numel()].view(p.shape).cpu()
        scale = max(float(p.grad.abs().max()), 1e-3 * gmax)
        err = float((got - p.grad).abs().max()) / scale
        checked += 1
        assert err < 2e-4, (name, err)
    assert checked == len(list(ref.parameters()))
    # one optimizer step on the flat buffers, then the refreshed weights must drive the next forward
    from audiodiffusion import training as T
    opt = T.AdamW(flat, lr=1e-3)
    ema = T.EMAModel(flat)
    clip = T.clip_grad_norm_(grads, 1.0)
    opt.step(grads, clip=clip, ema=ema, ema_decay=ema.next_decay())
    mine.refresh_weights()
    ref_opt = torch.optim.AdamW(ref.parameters(), lr=1e-3, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)
    torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
    ref_opt.step()
    ref_opt.zero_grad()
    loss2_ref = F.mse_loss(ref(x, ts)["sample"], tgt)
    loss2 = mine.train_step(x.to(dev), ts, tgt.to(dev))
    assert abs(float(loss2) - float(loss2_ref.detach())) <= 1e-4 * float(loss2_ref.detach())
    assert float(loss2) < float(loss)


# ---------------------------------------------------------------- data-parallel equivalence
def _ddp_worker(rank, world, port, q, overlap=False):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "audio-diffusion_amd"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      ADM_EMU_THREADS="2")
    import torch.distributed as dist
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from native_backend import select
    select("emu")
    from audiodiffusion import training as T
    from audiodiffusion.unet import UNet2DModel
    m = UNet2DModel(**TINY).init_random(0)
    flat, grads = m.enable_training()
    g = torch.Generator().manual_seed(1)
    x, tgt = torch.randn(4, 1, 16, 16, generator=g), torch.randn(4, 1, 16, 16, generator=g)
    ts = torch.tensor([5, 500, 999, 250])
    sl = slice(rank * 2, rank * 2 + 2)
    r = T.GradAllReducer(grads, bucket_mb=0.05)
    if overlap:                                   # all-reduce buckets are queued from inside the reverse pass (DDP overlap)
        r.attach(m)
    r.begin_step()
    m.train_step(x[sl].contiguous(), ts[sl], tgt[sl].contiguous())
    n_over = r.overlapped
    r.start(), r.finish()
    if overlap:
        assert n_over == len(r.bounds), (n_over, len(r.bounds))     # every bucket fired during the backward pass
        # a second step re-arms the hook; a no_sync micro-step must not touch the network
        r.begin_step()
        r.enabled = False
        m.train_step(x[sl].contiguous(), ts[sl], tgt[sl].contiguous())
        assert r.overlapped == 0 and not r.pending
        r.enabled = True
        r.begin_step()
        m.train_step(x[sl].contiguous(), ts[sl], tgt[sl].contiguous())
        assert r.overlapped == len(r.bounds)
        r.start(), r.finish()
    if rank == 0:
        q.put(grads.cpu().numpy().copy())  # by value: the producer may exit before the parent reads
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True], ids=["after-backward", "overlapped"])
def test_data_parallel_gradients_equal_full_batch(overlap):
    """2 ranks x (B/2) with the bucketed all-reduce == 1 rank x B, to fp32 tolerance (DDP semantics, train_unet.py:259);
    `overlapped`: the buckets are queued by the native reverse pass's bucket hook while it is still running."""
    import os
    import torch.multiprocessing as mp
    select("emu")
    from audiodiffusion.unet import UNet2DModel
    m = UNet2DModel(**TINY).init_random(0)
    flat, grads = m.enable_training()
    g = torch.Generator().manual_seed(1)
    x, tgt = torch.randn(4, 1, 16, 16, generator=g), torch.randn(4, 1, 16, 16, generator=g)
    m.train_step(x, torch.tensor([5, 500, 999, 250]), tgt)
    full = grads.clone()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + os.getpid() % 2000
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port + (7 if overlap else 0), q, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    got = torch.from_numpy(q.get(timeout=240))
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    assert float((got - full).abs().max()) <= 2e-5 * float(full.abs().max())


@pytest.mark.parametrize("backend", BACKENDS)
def test_gradient_accumulation_equals_one_large_batch(backend):
    """accelerator.accumulate (train_unet.py:252): k micro-batches with loss / k == one batch of k x the size."""
    dev = select(backend)
    from audiodiffusion import training as T
    from audiodiffusion.unet import UNet2DModel
    m = UNet2DModel(**TINY).init_random(0)
    flat, grads = m.enable_training()
    g = torch.Generator().manual_seed(5)
    x = torch.randn((4, 1, 16, 16), generator=g).to(dev)
    tgt = torch.randn((4, 1, 16, 16), generator=g).to(dev)
    ts = torch.tensor([3, 400, 700, 990])
    m.train_step(x, ts, tgt)
    full = grads.clone()
    acc = T.GradAccumulator(grads, 2)
    m.train_step(x[:2].contiguous(), ts[:2], tgt[:2].contiguous())
    assert acc.add() is False                     # micro-step: no sync, gradients parked in the accumulator
    m.train_step(x[2:].contiguous(), ts[2:], tgt[2:].contiguous())
    assert acc.add() is True
    scale = float(full.abs().max())
    assert float((grads - full).abs().max()) <= 2e-6 * scale
    # forced sync at the end of the dataloader keeps the 1/k scaling of the single accumulated micro-batch
    m.train_step(x[:2].contiguous(), ts[:2], tgt[:2].contiguous())
    half = grads.clone()
    assert acc.add(last_batch=True) is True
    assert float((grads - 0.5 * half).abs().max()) <= 1e-7 * scale
    # EMA on a micro-step: the same update the fused optimizer kernel applies
    ema = T.EMAModel(flat)
    ema.optimization_step = 10
    shadow0 = ema.shadow.clone()
    flat.add_(0.01)
    d = ema.step(flat)
    assert 0.0 < d < 1.0
    assert torch.allclose(ema.shadow, shadow0 - (1 - d) * (shadow0 - flat), atol=1e-7)


@pytest.mark.parametrize("backend", BACKENDS)
def test_partial_last_batch_runs_inside_the_full_batch_plan(backend):
    """`drop_last=False` (train_unet.py:181): the short last batch of an epoch must not re-plan the arena — it runs
    inside the full batch's plan: same workspace size before and after, gradients equal to those of a model that was planned at
    the short size from the start, also after optimizer steps + refresh_weights (the learned packing masks cover both sizes),
    and enable_training() called twice starts from loss scale 1 again."""
    dev = select(backend)
    from audiodiffusion import _native as N
    from audiodiffusion import training as T
    from audiodiffusion.unet import UNet2DModel
    g = torch.Generator().manual_seed(8)
    x = torch.randn((3, 1, 16, 16), generator=g).to(dev)
    tgt = torch.randn((3, 1, 16, 16), generator=g).to(dev)
    ts = torch.tensor([10, 500, 900])

    def fresh():
        m = UNet2DModel(**TINY).init_random(0)
        flat, grads = m.enable_training()
        return m, flat, grads, T.AdamW(flat, lr=1e-3)

    a, fa, ga, oa = fresh()           # full, partial, full, partial — one arena
    b, fb, gb, ob = fresh()           # every batch on its own (re-planned) size: partial batches only
    sizes = []
    for step in range(2):
        a.train_step(x, ts, tgt)
        sizes.append(N.lib().adm_unet_workspace_bytes(a._handle))
        oa.step(ga, clip=T.clip_grad_norm_(ga, 1.0)); a.refresh_weights()
        la = a.train_step(x[:1].contiguous(), ts[:1], tgt[:1].contiguous())
        sizes.append(N.lib().adm_unet_workspace_bytes(a._handle))
        # b follows a's parameters exactly, then takes the partial batch in a plan of its own size
        fb.copy_(fa)
        if step > 0:
            b.refresh_weights()       # (the first pass packs from the flat buffer itself)
        lb = b.train_step(x[:1].contiguous(), ts[:1], tgt[:1].contiguous())
        assert abs(float(la) - float(lb)) <= 1e-6 * abs(float(lb))
        assert float((ga - gb).abs().max()) <= 2e-6 * float(gb.abs().max())
        oa.step(ga, clip=T.clip_grad_norm_(ga, 1.0)); a.refresh_weights()
    assert len(set(sizes)) == 1 and sizes[0] > 0, sizes          # never re-planned
    a.train_step(x, ts, tgt, loss_scale=8.0)
    scaled = ga.clone()
    a.sync_state_dict_from_flat()
    a.enable_training()               # new native handle (same parameters): loss scale back to 1 on both sides
    _, ga2 = a.flat.data, a.flat_grads
    a.train_step(x, ts, tgt)
    assert float((scaled - 8.0 * ga2).abs().max()) <= 1e-5 * float(scaled.abs().max())
    a.train_step(x, ts, tgt, loss_scale=8.0)
    assert float((scaled - ga2).abs().max()) <= 1e-5 * float(scaled.abs().max())


# ---------------------------------------------------------------- --mixed_precision bf16 (train_unet.py:391-401, config 5)
BF16CFG = dict(sample_size=16, in_channels=1, out_channels=1, layers_per_block=1, block_out_channels=(128, 128),
               down_block_types=("DownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "UpBlock2D"))


@pytest.mark.parametrize("backend", BACKENDS)
def test_mixed_precision_bf16_training_step(backend):
    """bf16 MFMA operands / fp32 accumulation for the eligible 3x3 convolutions of forward, data gradient and weight
    gradient. Bars: (1) within bf16 tolerance of the fp32 autograd oracle, per parameter tensor; (2) no less accurate than
    the reference's own mixed-precision mode (torch.autocast(bf16) on the oracle, what accelerate applies); (3) the bf16
    kernels really ran (the result differs from the fp32 native path); (4) the loss still goes down after an optimizer
    step that re-rounds the bf16 filters from the fp32 masters."""
    dev = select(backend)
    from audiodiffusion import _native
    from audiodiffusion import training as T
    from audiodiffusion.unet import UNet2DModel
    torch.manual_seed(0)
    ref = OracleUNet(**BF16CFG)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "norm" in n:
                p.add_(0.2 * torch.randn_like(p))
    g = torch.Generator().manual_seed(1)
    x, tgt = torch.randn((2, 1, 16, 16), generator=g), torch.randn((2, 1, 16, 16), generator=g)
    ts = torch.tensor([5, 700])
    loss_ref = F.mse_loss(ref(x, ts)["sample"], tgt)
    loss_ref.backward()
    g32 = {n: p.grad.clone() for n, p in ref.named_parameters()}
    ref.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        loss_ac = F.mse_loss(ref(x, ts)["sample"].float(), tgt)
    loss_ac.backward()
    gac = {n: p.grad.clone() for n, p in ref.named_parameters()}
    ref.zero_grad()

    def native(mp):
        m = UNet2DModel(**BF16CFG).load_state_dict(ref.state_dict())
        flat, grads = m.enable_training(mixed_precision=mp)
        loss = float(m.train_step(x.to(dev), ts, tgt.to(dev)))
        return m, flat, grads, loss

    try:
        m, flat, grads, loss = native("bf16")
        gmax = max(float(v.abs().max()) for v in g32.values())

        def errs(get):
            """(global relative L2 error of the whole gradient, worst per-tensor max error relative to the larger of the
            tensor's own scale and a tenth of the global one — a lone scalar such as conv_out.bias is a cancelling sum and
            carries percent-level noise relative to itself under ANY bf16 scheme, the reference's autocast included)"""
            num = den = 0.0
            worst = 0.0
            for n, p in ref.named_parameters():
                d = get(n) - g32[n]
                num += float(d.double().pow(2).sum())
                den += float(g32[n].double().pow(2).sum())
                worst = max(worst, float(d.abs().max()) / max(float(g32[n].abs().max()), 0.1 * gmax))
            return (num / den) ** 0.5, worst

        mine = errs(lambda n: grads[m.flat.offsets[n][0]:m.flat.offsets[n][0] + g32[n].numel()].view(g32[n].shape).cpu())
        auto = errs(lambda n: gac[n].float())
        assert abs(loss - float(loss_ref.detach())) <= 5e-3 * float(loss_ref.detach()), (loss, float(loss_ref.detach()))
        assert mine[0] < 1.5e-2 and mine[1] < 2e-2, mine                                   # (1)
        assert mine[0] <= auto[0], (mine, auto)                                            # (2)
        assert mine[0] > 1e-4                                                              # (3) bf16 rounding is visible
        opt = T.AdamW(flat, lr=1e-4)        # Adam's first step moves every weight by lr: 1e-3 overshoots at this width in fp32 too
        opt.step(grads, clip=T.clip_grad_norm_(grads, 1.0))
        m.refresh_weights()
        loss2 = float(m.train_step(x.to(dev), ts, tgt.to(dev)))
        assert loss2 < loss                                                                # (4)
        # (5) refresh_weights re-packs only the weight images the training passes read (here: the bf16 ones); an inference call
        # through the SAME handle (evaluation samples, `--save_images_epochs`) runs fp32 and must first bring the fp32 /
        # Winograd images up to date: equal to a fresh inference model built from the current master weights
        opt.step(grads, clip=T.clip_grad_norm_(grads, 1.0))
        m.refresh_weights()                  # second refresh: the usage masks are known now, fp32 / Winograd images are skipped
        through_training_handle = m(x.to(dev), ts)["sample"].cpu()
        m.sync_state_dict_from_flat()
        fresh = UNet2DModel(**BF16CFG).load_state_dict(m.state_dict())
        want = fresh(x.to(dev), ts)["sample"].cpu()
        # (not bit-equal: the inference model takes its GroupNorm statistics from the convolutions' epilogues, the training
        # model from the read pass) — two optimizer steps moved the output by far more than the bar
        assert float((through_training_handle - want).abs().max()) <= 2e-5 * float(want.abs().max())
        before = UNet2DModel(**BF16CFG).load_state_dict(ref.state_dict())(x.to(dev), ts)["sample"].cpu()
        assert float((before - want).abs().max()) >= 2e-3 * float(want.abs().max())
        loss3 = float(m.train_step(x.to(dev), ts, tgt.to(dev)))                            # and training goes on unharmed
        assert loss3 < loss2
    finally:
        _native.check(_native.lib().adm_set_option(b"conv_bf16", 0))
    # the option is per training setup: a model enabled with "no" afterwards is fp32 again
    m32, _, grads32, loss32 = native("no")
    e32 = max(float((grads32[m32.flat.offsets[n][0]:m32.flat.offsets[n][0] + g32[n].numel()].view(g32[n].shape).cpu()
                     - g32[n]).abs().max()) / max(float(g32[n].abs().max()), 1e-2 * gmax) for n in g32)
    assert e32 < 2e-4 and abs(loss32 - float(loss_ref.detach())) <= 1e-5 * float(loss_ref.detach())


BLKCFG = dict(sample_size=32, in_channels=1, out_channels=1, layers_per_block=1, block_out_channels=(128, 128),
              down_block_types=("DownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "UpBlock2D"))


@pytest.mark.parametrize("backend", BACKENDS)
def test_mixed_precision_bf16_level3_blocked_operand_images(backend, monkeypatch):
    """Level 3 (round 4, k_conv_bf16b.hip): the 3x3 stride-1 convolutions of all three passes read blocked 16-bit operand images
    (activated input written once per layer, dy once per layer) through LDS-DMA.  32x32 resolution: the first level takes the
    32-pixel-row tiling, the 16x16 level (B = 2: one pair of images per tile) the narrow-row tiling with split K.  Bars: the toy model's (1) and (3)
    against fp32 autograd, no worse than torch.autocast, and the same gradient as level 2 up to accumulation order / rounding flips."""
    dev = select(backend)
    from audiodiffusion import _native
    from audiodiffusion.unet import UNet2DModel
    torch.manual_seed(0)
    ref = OracleUNet(**BLKCFG)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "norm" in n:
                p.add_(0.2 * torch.randn_like(p))
    g = torch.Generator().manual_seed(1)
    x, tgt = torch.randn((2, 1, 32, 32), generator=g), torch.randn((2, 1, 32, 32), generator=g)
    ts = torch.tensor([5, 700])
    loss_ref = F.mse_loss(ref(x, ts)["sample"], tgt)
    loss_ref.backward()
    g32 = torch.cat([p.grad.flatten() for _, p in ref.named_parameters()])
    ref.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        loss_ac = F.mse_loss(ref(x, ts)["sample"].float(), tgt)
    loss_ac.backward()
    gac = torch.cat([p.grad.float().flatten() for _, p in ref.named_parameters()])

    def native(level):
        monkeypatch.setenv("ADM_BF16_LEVEL", str(level))
        m = UNet2DModel(**BLKCFG).load_state_dict(ref.state_dict())
        _, grads = m.enable_training(mixed_precision="bf16")
        loss = float(m.train_step(x.to(dev), ts, tgt.to(dev)))
        flat = torch.cat([grads[m.flat.offsets[n][0]:m.flat.offsets[n][0] + p.numel()].cpu() for n, p in ref.named_parameters()])
        return loss, flat

    rel = lambda a, b: float((a - b).double().norm() / b.double().norm())  # noqa: E731
    try:
        l3, g3 = native(3)
        l2, g2 = native(2)
    finally:
        _native.check(_native.lib().adm_set_option(b"conv_bf16", 0))
    assert abs(l3 - float(loss_ref.detach())) <= 5e-3 * float(loss_ref.detach())
    e3, e2, eac = rel(g3, g32), rel(g2, g32), rel(gac, g32)
    assert 1e-4 < e3 < 1.5e-2, e3
    assert e3 <= eac, (e3, eac)
    # (level 3 also puts the stride-2 forward and weight gradient on 16-bit operands, which level 2 keeps fp32: 6e-3 apart, measured)
    assert rel(g3, g2) < 1e-2 and abs(e3 - e2) < 3e-3, (rel(g3, g2), e3, e2)
    assert float((g3 - g2).abs().max()) > 0          # another kernel family really ran
    # conv1's dy image written directly by the GroupNorm backward of its only reader vs the fp32 dx tensor + image pass: the same bits
    _native.check(_native.lib().adm_set_option(b"blk_direct_dy", 0))
    try:
        l3t, g3t = native(3)
    finally:
        _native.check(_native.lib().adm_set_option(b"blk_direct_dy", -1))
        _native.check(_native.lib().adm_set_option(b"conv_bf16", 0))
    assert l3t == l3 and torch.equal(g3, g3t)


UP8CFG = dict(sample_size=32, in_channels=1, out_channels=1, layers_per_block=1, block_out_channels=(128, 128, 128, 128),
              down_block_types=("DownBlock2D",) * 4, up_block_types=("UpBlock2D",) * 4)


@pytest.mark.parametrize("backend", BACKENDS)
def test_level3_upsample_onto_an_8x8_plane_stays_off_the_blocked_kernels(backend, monkeypatch):
    """at level 3 an Upsample2D conv whose OUTPUT plane is 8x8 (4x4 -> 8x8) passed the blocked kernels' shape test when
    B % 4 == 0, but they have no nearest-x2 variant for 8-pixel rows: `train_step` failed with 'conv_bf16b: no nearest-x2 variant for
    8-pixel rows' (sample_size 32, four levels of 128 channels, B = 4 — and the stock 256-model config at 64x64 / 32x32). The plan now
    leaves such a layer on the fp32 kernels; the step must run and meet the toy model's gradient bar."""
    dev = select(backend)
    from audiodiffusion import _native
    from audiodiffusion.unet import UNet2DModel
    monkeypatch.setenv("ADM_BF16_LEVEL", "3")
    torch.manual_seed(0)
    ref = OracleUNet(**UP8CFG)
    g = torch.Generator().manual_seed(3)
    x, tgt = torch.randn((4, 1, 32, 32), generator=g), torch.randn((4, 1, 32, 32), generator=g)
    ts = torch.tensor([5, 700, 33, 999])
    loss_ref = F.mse_loss(ref(x, ts)["sample"], tgt)
    loss_ref.backward()
    g32 = torch.cat([p.grad.flatten() for _, p in ref.named_parameters()])
    try:
        m = UNet2DModel(**UP8CFG).load_state_dict(ref.state_dict())
        _, grads = m.enable_training(mixed_precision="bf16")
        loss = float(m.train_step(x.to(dev), ts, tgt.to(dev)))
        flat = torch.cat([grads[m.flat.offsets[n][0]:m.flat.offsets[n][0] + p.numel()].cpu() for n, p in ref.named_parameters()])
    finally:
        _native.check(_native.lib().adm_set_option(b"conv_bf16", 0))
    assert abs(loss - float(loss_ref.detach())) <= 5e-3 * float(loss_ref.detach())
    e = float((flat - g32).double().norm() / g32.double().norm())
    assert 1e-4 < e < 1.5e-2, e


@pytest.mark.parametrize("backend", BACKENDS)
def test_level3_partial_batch_leaves_the_narrow_row_tilings(backend, monkeypatch):
    """The narrow-row tilings put 2 images of the 16x16 level side by side: a plan made for B = 4 takes them, the short last batch of an
    epoch (B = 3, `drop_last=False`, train_unet.py:181) cannot fill the tiles and must fall back INSIDE the same plan — forward, weight
    gradient, data gradient, and the GroupNorm backward that would otherwise write the producer's dy image directly — with up-to-date
    weight images (optimizer step in between), to the result of a model planned at B = 3 from the start."""
    dev = select(backend)
    from audiodiffusion import _native as N
    from audiodiffusion import training as T
    from audiodiffusion.unet import UNet2DModel
    monkeypatch.setenv("ADM_BF16_LEVEL", "3")
    g = torch.Generator().manual_seed(8)
    x = torch.randn((4, 1, 32, 32), generator=g).to(dev)
    tgt = torch.randn((4, 1, 32, 32), generator=g).to(dev)
    ts = torch.tensor([10, 500, 900, 77])
    try:
        a = UNet2DModel(**BLKCFG).init_random(0)
        fa, ga = a.enable_training(mixed_precision="bf16")
        oa = T.AdamW(fa, lr=1e-3)
        b = UNet2DModel(**BLKCFG).init_random(0)
        fb, gb = b.enable_training(mixed_precision="bf16")
        a.train_step(x, ts, tgt)
        ws = N.lib().adm_unet_workspace_bytes(a._handle)
        oa.step(ga, clip=T.clip_grad_norm_(ga, 1.0)); a.refresh_weights()
        la = a.train_step(x[:3].contiguous(), ts[:3], tgt[:3].contiguous())
        assert N.lib().adm_unet_workspace_bytes(a._handle) == ws
        fb.copy_(fa)
        lb = b.train_step(x[:3].contiguous(), ts[:3], tgt[:3].contiguous())
    finally:
        N.check(N.lib().adm_set_option(b"conv_bf16", 0))
    assert abs(float(la) - float(lb)) <= 1e-6 * abs(float(lb))
    # (bias gradients: channel sums of the dy image pass vs adm_chan_sums — another fp32 summation order)
    assert float((ga - gb).abs().max()) <= 1e-5 * float(gb.abs().max()), float((ga - gb).abs().max()) / float(gb.abs().max())


@pytest.mark.parametrize("backend", BACKENDS)
def test_mixed_precision_fp16_training_step_and_grad_scaler(backend):
    """`--mixed_precision fp16` (train_unet.py:391-395): the 16-bit-operand kernels on IEEE binary16 + GradScaler semantics.
    (1) with the loss scaled by 65536 the UN-scaled gradient is within fp16 tolerance of the fp32 autograd oracle (and closer
    than bf16's bar: 11 significand bits instead of 8); (2) the scale does not change the un-scaled result beyond rounding;
    (3) gradients too small for binary16 are flushed without the scale and survive with it; (4) an overflow is detected, the step skipped and the
    scale halved; (5) the scale doubles after `growth_interval` finite steps."""
    dev = select(backend)
    from audiodiffusion import training as T
    from audiodiffusion.unet import UNet2DModel
    torch.manual_seed(0)
    ref = OracleUNet(**BF16CFG)
    g = torch.Generator().manual_seed(1)
    x, tgt = torch.randn((2, 1, 16, 16), generator=g), torch.randn((2, 1, 16, 16), generator=g)
    ts = torch.tensor([5, 700])
    loss_ref = F.mse_loss(ref(x, ts)["sample"], tgt)
    loss_ref.backward()
    g32 = {n: p.grad.clone() for n, p in ref.named_parameters()}
    gnorm = float(torch.sqrt(sum(v.double().pow(2).sum() for v in g32.values())))

    m = UNet2DModel(**BF16CFG).load_state_dict(ref.state_dict())
    flat, grads = m.enable_training(mixed_precision="fp16")
    scaler = T.GradScaler(growth_interval=2)

    def rel_err(scale):
        loss = float(m.train_step(x.to(dev), ts, tgt.to(dev), loss_scale=scale))
        num = den = 0.0
        for n, p in ref.named_parameters():
            off = m.flat.offsets[n][0]
            d = grads[off:off + p.numel()].view(p.shape).cpu() / scale - g32[n]
            num += float(d.double().pow(2).sum()); den += float(g32[n].double().pow(2).sum())
        return loss, (num / den) ** 0.5

    loss, e_scaled = rel_err(scaler.get_scale())
    o_in, n_in0 = m.flat.offsets["conv_in.weight"][0], g32["conv_in.weight"].numel()
    w_in = g32["conv_in.weight"].flatten()                       # (3) below, read off this same step (scale 65536)
    fl_65536 = float(((grads[o_in:o_in + n_in0].cpu() / scaler.get_scale() - w_in).norm() / w_in.norm()))
    assert abs(loss - float(loss_ref.detach())) <= 2e-3 * float(loss_ref.detach())
    assert 1e-6 < e_scaled < 3e-3, e_scaled                                                  # (1) fp16 rounding visible, far inside bf16's 1.5e-2
    clip, found_inf = scaler.unscale_and_clip_(grads, 1.0)
    assert not found_inf and abs(float(clip[0]) - gnorm) <= 5e-3 * gnorm                      # the TRUE norm is reported
    assert abs(float(clip[1]) * scaler.get_scale() - min(1.0, 1.0 / (gnorm + 1e-6))) <= 5e-3   # coefficient = clip / scale
    _, e_one = rel_err(1.0)
    assert abs(e_one - e_scaled) < 3e-3                                                       # (2)
    # (3) binary16's range is why the scale exists: with the loss gradient scaled DOWN by 1e-7 (what small late-training
    # gradients look like) the first layer's gradient — which has come through every binary16 data-gradient convolution — is
    # flushed away, while at 65536 it is within a percent of fp32 autograd
    off, n_in = m.flat.offsets["conv_in.weight"][0], g32["conv_in.weight"].numel()
    want = g32["conv_in.weight"].flatten()

    def first_layer_err(scale, run=True):
        if run:
            m.train_step(x.to(dev), ts, tgt.to(dev), loss_scale=scale)
        return float(((grads[off:off + n_in].cpu() / scale - want).norm() / want.norm()))

    assert fl_65536 < 1e-2
    assert first_layer_err(1e-7) > 0.5
    # (4) overflow: a huge scale makes the operands infinite -> non-finite norm -> skipped step, scale halves
    m.train_step(x.to(dev), ts, tgt.to(dev), loss_scale=1e38)
    big = T.GradScaler(init_scale=1e38)
    _, inf = big.unscale_and_clip_(grads, 1.0)
    assert inf
    big.update(inf)
    assert big.step_was_skipped and big.get_scale() == 0.5e38
    # (5) growth
    s0 = scaler.get_scale()
    scaler.update(False); assert scaler.get_scale() == s0
    scaler.update(False); assert scaler.get_scale() == 2 * s0 and not scaler.step_was_skipped
    # an optimizer step with the scaled-gradient coefficient lowers the loss
    loss_a, _ = rel_err(s0)
    clip, _ = T.GradScaler(init_scale=s0).unscale_and_clip_(grads, 1.0)
    opt = T.AdamW(flat, lr=1e-4)
    opt.step(grads, clip=clip)
    m.refresh_weights()
    loss_b, _ = rel_err(s0)
    assert loss_b < loss_a
