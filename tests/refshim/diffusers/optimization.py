"""diffusers.optimization.get_scheduler (0.24): the two schedules the reference's --lr_scheduler flag is used with."""
import math

from torch.optim.lr_scheduler import LambdaLR


def get_scheduler(name, optimizer, num_warmup_steps=None, num_training_steps=None):
    if name == "constant":
        return LambdaLR(optimizer, lambda _: 1.0)
    if name != "cosine":
        raise NotImplementedError(name)

    def lr_lambda(step):                      # get_cosine_schedule_with_warmup, num_cycles = 0.5
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * 0.5 * 2.0 * progress)))
    return LambdaLR(optimizer, lr_lambda)
