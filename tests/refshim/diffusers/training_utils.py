"""diffusers.training_utils.EMAModel (0.24) as the reference's script drives it: constructed from the nn.Module with the
deprecated keywords `max_value` / `inv_gamma` / `power` (-> decay = max_value, warm-up schedule on), stepped with the module."""
import torch


class EMAModel:
    def __init__(self, parameters, decay=0.9999, min_decay=0.0, update_after_step=0, use_ema_warmup=False, inv_gamma=1.0,
                 power=2 / 3, **kwargs):
        if isinstance(parameters, torch.nn.Module):      # deprecated path: warm-up on
            parameters = parameters.parameters()
            use_ema_warmup = True
        if kwargs.get("max_value") is not None:
            decay = kwargs["max_value"]
        if kwargs.get("min_value") is not None:
            min_decay = kwargs["min_value"]
        self.shadow_params = [p.clone().detach() for p in parameters]
        self.decay, self.min_decay, self.update_after_step = decay, min_decay, update_after_step
        self.use_ema_warmup, self.inv_gamma, self.power = use_ema_warmup, inv_gamma, power
        self.optimization_step = 0
        self.cur_decay_value = None

    def get_decay(self, optimization_step):
        step = max(0, optimization_step - self.update_after_step - 1)
        if step <= 0:
            return 0.0
        if self.use_ema_warmup:
            cur = 1 - (1 + step / self.inv_gamma) ** -self.power
        else:
            cur = (1 + step) / (10 + step)
        return max(min(cur, self.decay), self.min_decay)

    @torch.no_grad()
    def step(self, parameters):
        if isinstance(parameters, torch.nn.Module):
            parameters = parameters.parameters()
        parameters = list(parameters)
        self.optimization_step += 1
        decay = self.get_decay(self.optimization_step)
        self.cur_decay_value = decay
        one_minus_decay = 1 - decay
        for s, p in zip(self.shadow_params, parameters):
            if p.requires_grad:
                s.sub_(one_minus_decay * (s - p))
            else:
                s.copy_(p)

    def copy_to(self, parameters):
        for s, p in zip(self.shadow_params, list(parameters)):
            p.data.copy_(s.to(p.device).data)
