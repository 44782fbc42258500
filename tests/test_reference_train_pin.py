"""Training step pinned to a run of the REFERENCE'S OWN scripts/train_unet.py (fixture
tests/golden/reference_train.npz, made by tests/golden/make_reference_train_golden.py: main(args) run as written for 2 epochs
x 3 steps on the hard-coded 113.67 M-parameter UNet at 32x32, over real accelerate / datasets and oracle-backed stand-ins for
diffusers / torchvision).  The fixture holds each step's clean batch, noise and timesteps, the logged loss / lr / ema_decay and
a per-tensor summary of the model saved after each epoch.
(a) wherever /root/reference exists, the generator reproduces the fixture (inputs exactly, floats to CPU-thread noise);
(b) CPU: a plain torch loop written from the script's text (autograd on the oracle UNet, torch.optim.AdamW, LambdaLR, EMA)
    reproduces the logged losses / learning rates and the saved weights;
(c) MI355X: the product's Trainer (scripts/train_unet.py: native forward+backward, fused clip + AdamW + EMA kernel) fed the
    recorded batches reproduces them within fp32 training tolerances."""
import importlib.util
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from native_backend import select

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, G)
from make_reference_train_golden import ARGS, REFERENCE, SEED, UNET_KW, state_digest, summarize  # noqa: E402

Z = np.load(os.path.join(G, "reference_train.npz"))
N_STEPS, PER_EPOCH = 6, 3


def _initial_oracle_unet():
    from oracle.unet import UNet2DModel
    torch.manual_seed(SEED)
    unet = UNet2DModel(**UNET_KW)
    assert state_digest(unet.state_dict()) == str(Z["init_sha256"]), "initial weights differ from the ones the reference run started from"
    return unet


def _check_summary(named, epoch, rel_l2, abs_head):
    names, rows = summarize(named)
    want_names, want = Z[f"epoch{epoch}:names"], Z[f"epoch{epoch}:summary"]
    assert list(names) == [str(n) for n in want_names]
    l2 = np.maximum(want[:, 1], 1e-12)
    assert (np.abs(rows[:, 1] - want[:, 1]) / l2).max() <= rel_l2                       # per-tensor L2 norm
    counts = np.array([p.numel() for _, p in named], dtype=np.float64)
    # per-tensor mean: within the L2 tolerance spread over the tensor
    assert (np.abs(rows[:, 0] - want[:, 0]) <= rel_l2 * l2 / np.sqrt(counts) + 1e-9).all()
    assert np.abs(rows[:, 2:] - want[:, 2:]).max() <= abs_head                          # first three entries
    return (np.abs(rows[:, 1] - want[:, 1]) / l2).max(), np.abs(rows[:, 2:] - want[:, 2:]).max()


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference tree only exists in the build container")
def test_the_committed_fixture_is_what_the_reference_script_produces(tmp_path):
    out = str(tmp_path / "again.npz")
    subprocess.run([sys.executable, os.path.join(G, "make_reference_train_golden.py"), out], check=True, capture_output=True,
                   timeout=900, cwd=str(tmp_path))
    again = np.load(out)
    assert sorted(again.files) == sorted(Z.files)
    for k in Z.files:
        if Z[k].dtype.kind == "f" and (k.endswith(":log") or k.endswith(":summary")):
            assert np.allclose(again[k], Z[k], rtol=1e-5, atol=1e-7), k        # multi-threaded CPU reductions: ~1e-8 run to run
        else:
            assert np.array_equal(again[k], Z[k]), k
    # what the script logs: lr after the step = cosine with 2 warm-up steps over 3*2//1 = 6 steps; ema_decay is the maximum
    lrs = [Z[f"step{i}:log"][1] for i in range(N_STEPS)]
    want = [1e-4 * (s / 2 if s < 2 else 0.5 * (1 + math.cos(math.pi * (s - 2) / 4))) for s in range(1, N_STEPS + 1)]
    assert np.allclose(lrs, want, rtol=1e-12, atol=1e-18) and all(Z[f"step{i}:log"][2] == 0.9999 for i in range(N_STEPS))
    assert [Z[f"step{i}:clean"].shape[0] for i in range(N_STEPS)] == [2, 2, 1, 2, 2, 1]       # drop_last=False


def test_plain_torch_loop_written_from_the_script_reproduces_the_reference_run():
    import torch.nn.functional as F
    from oracle.schedulers import DDPMScheduler
    unet = _initial_oracle_unet().train()
    sched = DDPMScheduler(num_train_timesteps=ARGS["num_train_steps"])
    opt = torch.optim.AdamW(unet.parameters(), lr=ARGS["learning_rate"], betas=(ARGS["adam_beta1"], ARGS["adam_beta2"]),
                            weight_decay=ARGS["adam_weight_decay"], eps=ARGS["adam_epsilon"])
    total = PER_EPOCH * ARGS["num_epochs"] // ARGS["gradient_accumulation_steps"]
    warm = ARGS["lr_warmup_steps"]
    lr_sched = torch.optim.lr_scheduler.LambdaLR(
        opt, lambda s: s / max(1, warm) if s < warm else max(0.0, 0.5 * (1 + math.cos(math.pi * (s - warm) / max(1, total - warm)))))
    shadow = [p.detach().clone() for p in unet.parameters()]
    for i in range(N_STEPS):
        clean, noise = torch.from_numpy(Z[f"step{i}:clean"]), torch.from_numpy(Z[f"step{i}:noise"])
        t = torch.from_numpy(Z[f"step{i}:timesteps"])
        loss = F.mse_loss(unet(sched.add_noise(clean, noise, t), t)["sample"], noise)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(unet.parameters(), 1.0)
        opt.step()
        lr_sched.step()
        step = i + 1                                               # EMAModel.step: decay = min(1 - (1 + (step-1)/inv_gamma)^-power, max)
        decay = 0.0 if step - 1 <= 0 else min(1 - (1 + (step - 1) / ARGS["ema_inv_gamma"]) ** -ARGS["ema_power"], ARGS["ema_max_decay"])
        with torch.no_grad():
            for s, p in zip(shadow, unet.parameters()):
                s.sub_((1 - decay) * (s - p))
        opt.zero_grad()
        want = Z[f"step{i}:log"]
        assert abs(loss.item() - want[0]) <= 2e-5 * want[0], (i, loss.item(), want[0])
        assert abs(lr_sched.get_last_lr()[0] - want[1]) <= 1e-12 and want[3] == step
        if (i + 1) % PER_EPOCH == 0:                               # save epoch: the EMA weights go INTO the live model (:292-294)
            with torch.no_grad():
                for s, p in zip(shadow, unet.parameters()):
                    p.copy_(s)
            _check_summary(list(unet.named_parameters()), (i + 1) // PER_EPOCH - 1, rel_l2=2e-6, abs_head=2e-6)


@pytest.mark.gpu
def test_product_trainer_reproduces_the_reference_run():
    dev = select("hip")
    from audiodiffusion import DDPMScheduler, UNet2DModel
    spec = importlib.util.spec_from_file_location("adm_train_unet_pin", os.path.join(ROOT, "audio-diffusion_amd", "scripts", "train_unet.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    flags = []
    for k in ("train_batch_size", "num_epochs", "gradient_accumulation_steps", "learning_rate", "lr_scheduler", "lr_warmup_steps",
              "adam_beta1", "adam_beta2", "adam_weight_decay", "adam_epsilon", "ema_inv_gamma", "ema_power", "ema_max_decay",
              "mixed_precision", "num_train_steps", "scheduler"):
        flags += [f"--{k}", str(ARGS[k])]
    args = mod.parse_args(flags)
    assert args.use_ema
    oracle_unet = _initial_oracle_unet()
    model = UNet2DModel(**UNET_KW).load_state_dict(oracle_unet.state_dict())
    names = [n for n, _ in oracle_unet.named_parameters()]
    tr = mod.Trainer(args, model, UNET_KW["sample_size"], PER_EPOCH, 1, dev)
    sched = DDPMScheduler(num_train_timesteps=ARGS["num_train_steps"])
    worst = 0.0
    for i in range(N_STEPS):
        clean, noise = torch.from_numpy(Z[f"step{i}:clean"]).to(dev), torch.from_numpy(Z[f"step{i}:noise"]).to(dev)
        loss = float(tr.step(sched, clean, noise, torch.from_numpy(Z[f"step{i}:timesteps"]), last_batch=((i + 1) % PER_EPOCH == 0)))
        want = Z[f"step{i}:log"]
        worst = max(worst, abs(loss - want[0]) / want[0])
        assert abs(loss - want[0]) <= 2e-5 * want[0], (i, loss, want[0])      # measured on MI355X: 1.2e-7
        assert abs(tr.lr_scheduler.get_last_lr()[0] - want[1]) <= 1e-12
        if (i + 1) % PER_EPOCH == 0:
            tr.copy_ema_into_model()
            sd = model.state_dict()
            # measured on MI355X: per-tensor L2 within 2.9e-6, entries within 1.2e-8 (AdamW's first updates are +-lr whatever
            # the gradient's size, so an entry whose gradient is rounding noise could in principle move the other way)
            got = _check_summary([(n, sd[n].cpu()) for n in names], (i + 1) // PER_EPOCH - 1, rel_l2=5e-5, abs_head=5e-6)
            print(f"epoch {(i + 1) // PER_EPOCH - 1}: worst per-tensor L2 difference {got[0]:.2e} (relative), worst entry difference {got[1]:.2e}")
    print(f"worst relative loss difference vs the reference run: {worst:.2e}")
