"""`models.t5` — drop-in for the reference module of the same name (models/t5.py:37-360).

`T5ForConditionalGeneration(config)` keeps the reference's constructor, `forward(inputs, labels)
-> lm_logits`, `generate(inputs, max_length)` and state-dict keys; the stack arithmetic runs in the
gfx950 kernels behind `mrmt3.engine` instead of HF `T5Block` on stock PyTorch ops.
"""
from mrmt3.module import MT3Module


class T5ForConditionalGeneration(MT3Module):
    VARIANT = "t5"

    def __init__(self, config, compute_dtype=None):
        import torch
        super().__init__(config, compute_dtype=compute_dtype or torch.bfloat16)
