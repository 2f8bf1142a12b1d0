"""`utils` — the LR schedule the reference's tasks use (utils.py:25-61)."""
import math

from torch.optim.lr_scheduler import LambdaLR


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


def remove_state_dict_prefix(state_dict, prefix="model."):
    """Lightning checkpoints prefix every key with `model.` (train.py:109-115)."""
    return {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in state_dict.items()}
