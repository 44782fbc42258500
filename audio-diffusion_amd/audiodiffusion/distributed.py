"""Batch-sharded sampling across the GPUs of one node (BASELINE.json config 3).

Samples never interact (convolutions, GroupNorm statistics, attention and the scheduler are per-sample), so the path
shards by rows with NO collective inside the denoising loop: one process per GPU (`torch.distributed`, backend "nccl" ==
RCCL over xGMI on the MI355X node, "gloo" in CPU tests), identical weights on every rank, the GLOBAL noise batch drawn from
one seed and row-sliced per rank, one `all_gather` of the uint8 images at the end. The result is byte-identical for any
world size. (The reference samples single-process only; this is the new capability the north_star names.)
"""
import torch
import torch.distributed as dist


def shard_bounds(global_batch, world, rank):
    per = (global_batch + world - 1) // world
    lo = min(rank * per, global_batch)
    return lo, min(lo + per, global_batch)


def global_noise(shape, seed, steps_with_noise=0):
    """Initial latent (and, for DDPM / eta>0, per-step noise) for the GLOBAL batch from one CPU generator."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g)
    sn = [torch.randn(shape, generator=g) for _ in range(steps_with_noise)]
    return x, sn


@torch.no_grad()
def sample_sharded(pipe, global_batch, steps=None, seed=42, eta=0.0, gather=True, group=None):
    """Returns (images_u8, local_slice): uint8 tensor (global_batch, H, W) on every rank when `gather`, else the
    local shard; `local_slice` = (lo, hi) rows owned by this rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    steps = steps or pipe.get_default_steps()
    pipe.scheduler.set_timesteps(steps)
    ss = pipe.unet.sample_size
    H, W = (ss, ss) if isinstance(ss, int) else ss
    rows = pipe.scheduler.coef_rows(eta)
    n_noise = sum(1 for r in rows if r["k_noise"] != 0.0)
    x, sn = global_noise((global_batch, pipe.unet.in_channels, H, W), seed, n_noise)
    lo, hi = shard_bounds(global_batch, world, rank)
    dev = pipe.device
    step_noise = None
    if n_noise:
        it = iter(sn)
        step_noise = [next(it)[lo:hi].to(dev) if r["k_noise"] != 0.0 else None for r in rows]
    if hi > lo:
        _, u8 = pipe._denoise(x[lo:hi].contiguous().to(dev), 0, eta, None, None, 0, 0, step_noise=step_noise)
        u8 = u8.reshape(hi - lo, H, W)
    else:      # more ranks than rows (global_batch < world * per): this rank owns nothing, and still takes part in the gather
        u8 = torch.zeros((0, H, W), dtype=torch.uint8, device=dev)
    if not gather or not dist.is_initialized():      # a 1-rank group still goes through the collective (RCCL shake-out)
        return u8, (lo, hi)
    per = (global_batch + world - 1) // world
    pad = torch.zeros((per, H, W), dtype=torch.uint8, device=dev)
    pad[: hi - lo] = u8
    out = torch.empty((world * per, H, W), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, pad, group=group)
    return out[:global_batch], (lo, hi)
