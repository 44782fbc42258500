"""Optimizer side of `scripts/train_unet.py` on flat device buffers.

Reference call sites: `train_unet.py:166-190` (AdamW lr 1e-4, betas (0.95, 0.999), wd 1e-6, eps 1e-8; cosine schedule
with 500 warm-up steps; `EMAModel(model, inv_gamma=1.0, power=3/4, max_value=0.9999)`), `:250` add_noise
(`schedulers.add_noise`), `:258` `F.mse_loss`, `:261-267` clip_grad_norm_(1.0) / optimizer.step / lr_scheduler.step /
ema_model.step, and the data-parallel gradient all-reduce that `accelerator.backward` performs through DDP (`:259`,
`config/accelerate_multi_gpu.yaml:3`).

Every per-parameter pass is ONE fused HIP kernel over a flat fp32 buffer (csrc/k_train.hip) instead of ~450 per-tensor
launches; loss / norm / clip scalars stay on the device (no host sync inside a step). The gradient all-reduce uses
`torch.distributed` (backend "nccl" == RCCL over xGMI on the MI355X node) in ~25 MB buckets of the flat gradient buffer.
The kernels that fill the gradient buffer are `adm_unet_forward_backward` (csrc/net_exec.hip: run_backward), driven by
`UNet2DModel.train_step`. No torch arithmetic op touches parameters, gradients or optimizer state: the remaining
elementwise passes (micro-step accumulation, 1/world, EMA on its own) go through `adm_flat_op`.
"""
import ctypes as C
import math

import torch
import torch.distributed as dist

from . import _native as N


class FlatBuffer:
    """name -> view bookkeeping over one contiguous fp32 device allocation (parameters, grads, moments, EMA)."""

    def __init__(self, specs, device):
        self.offsets, off = {}, 0
        for name, shape in specs:
            n = math.prod(shape)
            self.offsets[name] = (off, tuple(shape))
            off += (n + 3) // 4 * 4  # keep every tensor 16-byte aligned for float4 access
        self.numel = off
        self.data = torch.zeros(off, dtype=torch.float32, device=device)

    def view(self, name):
        off, shape = self.offsets[name]
        return self.data[off:off + math.prod(shape)].view(shape)

    def load(self, state_dict):
        for k, v in state_dict.items():
            self.view(k).copy_(v)
        return self


FLAT_ADD, FLAT_SCALE_FROM, FLAT_DIV, FLAT_EMA = 0, 1, 2, 3


def flat_op(y, x, op, a=0.0):
    """y (op)= x on flat fp32 buffers (csrc/k_train.hip: flat_op_kernel)."""
    N.check(N.lib().adm_flat_op(N.ptr(y), N.ptr(x), y.numel(), op, float(a), N.stream_for(y)))
    return y


def mse_loss(pred, target, want_grad=True):
    """F.mse_loss (train_unet.py:258). Returns (loss: 0-d device tensor, dloss/dpred or None)."""
    pred, target = pred.contiguous(), target.contiguous()
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    grad = torch.empty_like(pred) if want_grad else None
    scratch = torch.zeros(1, dtype=torch.float64, device=pred.device)
    N.check(N.lib().adm_mse_loss(N.ptr(pred), N.ptr(target), pred.numel(), N.ptr(loss), N.ptr(grad), N.ptr(scratch),
                                 N.stream_for(pred)))
    return loss[0], grad


def clip_grad_norm_(flat_grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ over the flat gradient buffer (train_unet.py:261-262). Returns a device tensor
    [total_norm, clip_coef]; the scaling itself is fused into `AdamW.step(clip=...)`."""
    out = torch.empty(2, dtype=torch.float32, device=flat_grads.device)
    scratch = torch.zeros(1, dtype=torch.float64, device=flat_grads.device)
    N.check(N.lib().adm_grad_norm_clip(N.ptr(flat_grads), flat_grads.numel(), float(max_norm), N.ptr(out), N.ptr(scratch),
                                       N.stream_for(flat_grads)))
    return out


class GradScaler:
    """torch.cuda.amp.GradScaler as accelerate drives it for `--mixed_precision fp16` (train_unet.py:391-395): dynamic loss
    scale (init 65536, x2 after `growth_interval` = 2000 finite steps, x0.5 on an overflow); un-scaling and clipping are ONE
    pass over the flat gradient buffer (`adm_grad_norm_clip_scaled`); an overflow (non-finite gradient norm) skips the
    optimizer step. Reading the norm back is the one host sync of an fp16 step (GradScaler's `found_inf.item()` likewise)."""

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.scale, self.growth_factor, self.backoff_factor = float(init_scale), growth_factor, backoff_factor
        self.growth_interval, self._good = growth_interval, 0
        self.step_was_skipped = False

    def get_scale(self):
        return self.scale

    def unscale_and_clip_(self, flat_grads, max_norm):
        """-> (norm_clip device tensor [true norm, coefficient incl. 1/scale] for AdamW.step(clip=...), found_inf: bool)."""
        out = torch.empty(2, dtype=torch.float32, device=flat_grads.device)
        scratch = torch.zeros(1, dtype=torch.float64, device=flat_grads.device)
        N.check(N.lib().adm_grad_norm_clip_scaled(N.ptr(flat_grads), flat_grads.numel(), float(max_norm), 1.0 / self.scale,
                                                  N.ptr(out), N.ptr(scratch), N.stream_for(flat_grads)))
        found_inf = not math.isfinite(float(out[0]))
        return out, found_inf

    def update(self, found_inf):
        """scaler.update(): back off on an overflow, grow after `growth_interval` consecutive finite steps."""
        self.step_was_skipped = bool(found_inf)
        if found_inf:
            self.scale *= self.backoff_factor
            self._good = 0
        else:
            self._good += 1
            if self._good >= self.growth_interval:
                self.scale *= self.growth_factor
                self._good = 0


class EMAModel:
    """diffusers==0.24.0 training_utils.EMAModel as the reference constructs it (train_unet.py:185-190): passing a
    module enables the warm-up schedule decay = min(1 - (1 + step/inv_gamma)^-power, max_value)."""

    def __init__(self, flat_params, inv_gamma=1.0, power=3 / 4, max_value=0.9999, min_decay=0.0, update_after_step=0):
        self.shadow = flat_params.clone()
        self.inv_gamma, self.power, self.decay, self.min_decay = inv_gamma, power, max_value, min_decay
        self.update_after_step = update_after_step
        self.optimization_step = 0
        self.cur_decay_value = 0.0

    def get_decay(self, optimization_step):
        step = max(0, optimization_step - self.update_after_step - 1)
        if step <= 0:
            return 0.0
        cur = 1 - (1 + step / self.inv_gamma) ** -self.power
        return max(min(cur, self.decay), self.min_decay)

    def next_decay(self):
        """Advance the step counter and return the decay `AdamW.step(ema=..., ema_decay=...)` must apply."""
        self.optimization_step += 1
        self.cur_decay_value = self.get_decay(self.optimization_step)
        return self.cur_decay_value

    def copy_to(self, flat_params):
        flat_params.copy_(self.shadow)

    def step(self, flat_params):
        """EMAModel.step on its own (the fused optimizer kernel applies the same update together with AdamW): used on
        gradient-accumulation micro-steps, where the reference still calls `ema_model.step(model)` (train_unet.py:265-266)
        although the optimizer does not move the parameters."""
        d = self.next_decay()
        flat_op(self.shadow, flat_params, FLAT_EMA, 1.0 - d)
        return d


class GradAccumulator:
    """`accelerator.accumulate(model)` (train_unet.py:252) for the flat gradient buffer: `adm_unet_forward_backward`
    overwrites `flat_grads` on every call, so micro-step gradients are summed here; on the synchronising step the
    sum / k (accelerate scales each micro-loss by 1 / gradient_accumulation_steps) goes back into `flat_grads` for the
    all-reduce, clipping and the optimizer. A sync is also forced at the end of the dataloader (accelerate's
    `sync_with_dataloader` default)."""

    def __init__(self, flat_grads, steps):
        self.grads, self.steps = flat_grads, int(steps)
        self.acc = torch.zeros_like(flat_grads) if self.steps > 1 else None
        self.count = 0

    def add(self, last_batch=False):
        """Call after each forward/backward; returns True when the optimizer should step (`sync_gradients`)."""
        if self.steps <= 1:
            return True
        flat_op(self.acc, self.grads, FLAT_ADD)
        self.count += 1
        if self.count % self.steps != 0 and not last_batch:
            return False
        flat_op(self.grads, self.acc, FLAT_SCALE_FROM, 1.0 / self.steps)
        self.acc.zero_()
        self.count = 0
        return True


class AdamW:
    """torch.optim.AdamW over one flat parameter buffer; optional fused clip factor and EMA update."""

    def __init__(self, flat_params, lr=1e-4, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8):
        self.params = flat_params
        self.lr, self.betas, self.weight_decay, self.eps = lr, betas, weight_decay, eps
        self.exp_avg = torch.zeros_like(flat_params)
        self.exp_avg_sq = torch.zeros_like(flat_params)
        self.step_count = 0

    def step(self, flat_grads, clip=None, ema=None, ema_decay=0.0):
        self.step_count += 1
        clip_ptr = None if clip is None else C.c_void_p(clip.data_ptr() + 4)  # [norm, coef] -> coef
        N.check(N.lib().adm_adamw_ema_step(
            N.ptr(self.params), N.ptr(flat_grads), N.ptr(self.exp_avg), N.ptr(self.exp_avg_sq),
            N.ptr(ema.shadow) if ema is not None else None, self.params.numel(), float(self.lr), float(self.betas[0]),
            float(self.betas[1]), float(self.eps), float(self.weight_decay), self.step_count, clip_ptr, float(ema_decay),
            N.stream_for(self.params)))


def get_cosine_schedule_with_warmup(num_warmup_steps, num_training_steps, num_cycles=0.5):
    """diffusers.optimization.get_scheduler("cosine", ...) multiplier (train_unet.py:174-179)."""
    def lr_lambda(current_step):
        if current_step < num_warmup_steps:
            return float(current_step) / float(max(1, num_warmup_steps))
        progress = float(current_step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))
    return lr_lambda


class LambdaLR:
    def __init__(self, optimizer, lr_lambda):
        self.opt, self.fn, self.base_lr, self.last_epoch = optimizer, lr_lambda, optimizer.lr, 0
        self.opt.lr = self.base_lr * self.fn(0)

    def step(self):
        self.last_epoch += 1
        self.opt.lr = self.base_lr * self.fn(self.last_epoch)

    def get_last_lr(self):
        return [self.opt.lr]


class GradAllReducer:
    """DDP-style gradient averaging of the flat gradient buffer: ~25 MB buckets (DDP's default bucket_cap_mb), each an
    async all-reduce so buckets pipeline over the xGMI links; `finish()` waits and applies the 1/world scaling that DDP
    applies. No collective is issued under gradient accumulation micro-steps (`no_sync`, train_unet.py:252)."""

    def __init__(self, flat_grads, bucket_mb=25, group=None, force=False):
        self.g, self.group = flat_grads, group
        self.force = bool(force)     # issue the collectives even in a 1-rank group (single-GPU shake-out of the RCCL path)
        per = max(1, int(bucket_mb * (1 << 20)) // 4)
        self.bounds = [(i, min(i + per, flat_grads.numel())) for i in range(0, flat_grads.numel(), per)]
        self.pending = []

        self.launched = set()
        self.enabled = True          # set False on gradient-accumulation micro-steps (DDP no_sync)
        self.overlapped = 0          # buckets whose all-reduce was queued from inside the backward pass (last step)
        self._cb = None

    def _active(self):
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.force)

    def _launch(self, b):
        lo, hi = self.bounds[b]
        self.launched.add(b)
        self.pending.append(dist.all_reduce(self.g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def attach(self, model):
        """Overlap with the backward pass, as DDP does: the native reverse pass calls back as soon as the last kernel
        writing into a bucket is enqueued; the bucket's asynchronous all-reduce is queued right then (RCCL orders it after
        the work already on the current stream) and runs while the remaining layers are still being differentiated."""
        bounds = (C.c_long * (len(self.bounds) + 1))(*([lo for lo, _ in self.bounds] + [self.bounds[-1][1]]))

        def on_bucket(_user, b):
            if self.enabled and self._active():
                self._launch(b)
                self.overlapped += 1

        self._cb = N.BUCKET_FN(on_bucket)            # keep the trampoline alive as long as the hook is installed
        N.check(N.lib().adm_unet_set_grad_bucket_hook(model._handle, len(self.bounds), bounds,
                                                      C.cast(self._cb, C.c_void_p), None))
        return self

    def start(self):
        """Queue every bucket the backward pass has not queued already (all of them without `attach`)."""
        if not self._active():
            return
        for b in range(len(self.bounds)):
            if b not in self.launched:
                self._launch(b)

    def finish(self):
        if not self.pending:
            self.launched = set()
            return
        for w in self.pending:
            w.wait()
        self.pending = []
        self.launched = set()
        flat_op(self.g, None, FLAT_DIV, float(dist.get_world_size(self.group)))

    def begin_step(self):
        """Call before the forward/backward of a synchronising step."""
        self.overlapped = 0
