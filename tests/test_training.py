"""Optimizer-side parity of the training step vs torch's own AdamW / clip / mse_loss and the
diffusers formulas restated here; gradient all-reduce over gloo with world_size 2."""
import math
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from native_backend import BACKENDS, select

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("backend", BACKENDS)
def test_mse_and_clip(backend):
    dev = select(backend)
    from audiodiffusion import training as T
    g = torch.Generator().manual_seed(0)
    a, b = torch.randn(3, 1, 16, 16, generator=g), torch.randn(3, 1, 16, 16, generator=g)
    loss, grad = T.mse_loss(a.to(dev), b.to(dev))
    ar = a.clone().requires_grad_(True)
    lr_ = F.mse_loss(ar, b)
    lr_.backward()
    assert abs(float(loss) - float(lr_.detach())) <= 1e-6 * float(lr_.detach())
    assert torch.allclose(grad.cpu(), ar.grad, atol=1e-8, rtol=1e-6)
    gflat = torch.randn(10007, generator=g) * 3
    out = T.clip_grad_norm_(gflat.to(dev), 1.0).cpu()
    p = torch.nn.Parameter(torch.zeros(10007))
    p.grad = gflat.clone()
    ref_norm = torch.nn.utils.clip_grad_norm_([p], 1.0)
    assert abs(float(out[0]) - float(ref_norm)) <= 1e-5 * float(ref_norm)
    assert abs(float(out[1]) - min(1.0, 1.0 / (float(ref_norm) + 1e-6))) <= 1e-6


@pytest.mark.parametrize("backend", BACKENDS)
def test_adamw_ema_matches_torch(backend):
    dev = select(backend)
    from audiodiffusion import training as T
    g = torch.Generator().manual_seed(1)
    n = 4099
    p0 = torch.randn(n, generator=g)
    ref_p = torch.nn.Parameter(p0.clone())
    ref_opt = torch.optim.AdamW([ref_p], lr=1e-2, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)
    mine_p = p0.clone().to(dev)
    opt = T.AdamW(mine_p, lr=1e-2, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)
    ema = T.EMAModel(mine_p, inv_gamma=1.0, power=3 / 4, max_value=0.9999)
    ref_shadow = p0.clone()
    sched = T.LambdaLR(opt, T.get_cosine_schedule_with_warmup(3, 10))
    ref_sched = torch.optim.lr_scheduler.LambdaLR(ref_opt, T.get_cosine_schedule_with_warmup(3, 10))
    for step in range(1, 8):
        grad = torch.randn(n, generator=g) * (5.0 if step == 2 else 0.3)
        ref_p.grad = grad.clone()
        torch.nn.utils.clip_grad_norm_([ref_p], 1.0)
        ref_opt.step(), ref_sched.step()
        # diffusers EMAModel.step with use_ema_warmup (reference passes a module): oracle formula
        s = max(0, step - 1)
        decay = 0.0 if s <= 0 else min(1 - (1 + s / 1.0) ** -0.75, 0.9999)
        ref_shadow -= (1 - decay) * (ref_shadow - ref_p.detach())
        gd = grad.to(dev)
        clip = T.clip_grad_norm_(gd, 1.0)
        d = ema.next_decay()
        assert abs(d - decay) < 1e-12
        opt.step(gd, clip=clip, ema=ema, ema_decay=d)
        sched.step()
        assert abs(opt.lr - ref_opt.param_groups[0]["lr"]) < 1e-12
        assert torch.allclose(mine_p.cpu(), ref_p.detach(), rtol=2e-6, atol=2e-7), step
        assert torch.allclose(ema.shadow.cpu(), ref_shadow, rtol=2e-6, atol=2e-7), step


def test_cosine_schedule_values():
    from audiodiffusion.training import get_cosine_schedule_with_warmup
    f = get_cosine_schedule_with_warmup(500, 10000)
    assert f(0) == 0.0 and f(250) == 0.5 and f(500) == 1.0
    assert abs(f(5250) - 0.5) < 1e-12 and f(10000) < 1e-12


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "audio-diffusion_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from native_backend import select
    select("emu")                 # finish() applies the 1/world scaling with the native flat-op kernel
    from audiodiffusion.training import GradAllReducer
    g = torch.arange(100000, dtype=torch.float32) * (rank + 1)
    r = GradAllReducer(g, bucket_mb=1 / 16)  # 16384-float buckets -> several buckets in flight
    r.start(), r.finish()
    if rank == 0:
        q.put(g.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_grad_allreduce_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = torch.from_numpy(q.get(timeout=120))
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert torch.allclose(out, torch.arange(100000, dtype=torch.float32) * 1.5)


@pytest.mark.parametrize("backend", BACKENDS)
def test_flat_ops_match_the_torch_expressions_they_replace(backend):
    """adm_flat_op vs the four torch expressions of the reference's optimizer side (accumulate / mean / 1-over-world / EMA)."""
    dev = select(backend)
    from audiodiffusion import training as T
    g = torch.Generator().manual_seed(3)
    for n in (4, 4099, 65536 + 4):                      # buffers are padded to float4 by FlatBuffer; tails still covered
        y0, x = torch.randn(n, generator=g), torch.randn(n, generator=g)
        for op, a, ref in ((T.FLAT_ADD, 0.0, lambda y: y + x), (T.FLAT_SCALE_FROM, 1.0 / 3, lambda y: x * (1.0 / 3)),
                           (T.FLAT_DIV, 3.0, lambda y: y / 3.0), (T.FLAT_EMA, 1 - 0.9, lambda y: y - (y - x) * (1 - 0.9))):
            y = y0.clone().to(dev)
            T.flat_op(y, None if op == T.FLAT_DIV else x.to(dev), op, a)
            want = ref(y0.clone())
            # one fp32 ulp of the operands: the device contracts y - a*(y - x) into an fma, torch rounds the product first
            tol = 2e-7 * float(torch.maximum(y0.abs(), x.abs()).max())
            assert float((y.cpu() - want).abs().max()) <= tol, (n, op, float((y.cpu() - want).abs().max()))
