"""`utils` — the reference's utils.py surface: the cosine-with-warm-up schedule its tasks use (:25-61) and the
small helpers next to it (Noam schedule :7-22, warm-up-only schedule :64-71, result dir :74-83, state-dict prefix
removal :86-90)."""
import math
from collections import OrderedDict

import torch
from torch.optim.lr_scheduler import LambdaLR


class NoamScheduler(torch.optim.lr_scheduler._LRScheduler):
    """lr = 0.002 * sqrt(d_model) * min(s^-0.5, s * warmup^-1.5) with s = last_epoch + 2 (utils.py:7-18)."""

    def __init__(self, optimizer, warmup_steps, model_dim, last_epoch=-1):
        self.warmup_steps = warmup_steps
        self.model_dim = model_dim
        super().__init__(optimizer, last_epoch)

    def get_lr(self, epoch=None):
        step = self.last_epoch + 2
        lr = 0.002 * self.model_dim ** 0.5 * min(step ** -0.5, step * self.warmup_steps ** -1.5)
        return [lr for _ in self.optimizer.param_groups]


def get_noam_scheduler(optimizer, warmup_steps, model_dim):
    return NoamScheduler(optimizer, warmup_steps, model_dim)


def cosine_warmup_lambda(num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5,
                         min_lr: float = 2e-5):
    """Multiplier applied to the optimizer's base lr.  As in the reference, `min_lr` is a floor on
    this MULTIPLIER (not on the learning rate)."""

    def lr_lambda(current_step):
        if current_step < num_warmup_steps:
            return float(current_step) / float(max(1, num_warmup_steps))
        progress = float(current_step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
        return max(min_lr, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))

    return lr_lambda


def get_cosine_schedule_with_warmup(optimizer, num_warmup_steps: int, num_training_steps: int,
                                    num_cycles: float = 0.5, last_epoch: int = -1, min_lr: float = 2e-5):
    return LambdaLR(optimizer, cosine_warmup_lambda(num_warmup_steps, num_training_steps, num_cycles, min_lr),
                    last_epoch)


def get_mt3_optimizer(optimizer, num_warmup_steps: int, last_epoch: int = -1):
    """Linear warm-up to the base lr, then constant (utils.py:64-71; despite the name it returns the scheduler)."""
    return LambdaLR(optimizer, lambda step: min(1, step / num_warmup_steps), last_epoch)


def get_result_dir(lightning_logs_dir="results"):
    """utils.py:74-83 (the experiment number it computes is not used there either)."""
    return f"./{lightning_logs_dir}"


def remove_state_dict_prefix(state_dict, prefix="module."):
    """utils.py:86-90: `prefix` is removed wherever it occurs in a key (str.replace), default `module.`.  Lightning's
    `model.` prefix is handled by `mrmt3.checkpoint.strip_prefix` (leading prefix only)."""
    out = OrderedDict()
    for k, v in state_dict.items():
        out[k.replace(prefix, "")] = v
    return out
