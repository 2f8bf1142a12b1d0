"""`models.t5_segmem` — drop-in for the reference's V1 segment-memory model
(models/t5_segmem.py:38-170): memory of the previous batch row prepended to the decoder input
embeddings.  `generate` is the plain batched greedy decode (`:254-311`, the memory is not used);
`generate_2` is the sequential memory-prefixed decode (`:172-252`)."""
import torch

from mrmt3.module import MT3Module


class T5SegMem(MT3Module):
    VARIANT = "segmem_v1"

    def __init__(self, config, segmem_num_layers: int = 1, segmem_length: int = 64, compute_dtype=None):
        super().__init__(config, segmem_num_layers=segmem_num_layers, segmem_length=segmem_length,
                         compute_dtype=compute_dtype or torch.bfloat16)

    def generate_2(self, inputs, max_length=1024, output_hidden_states=False, **kwargs):
        from mrmt3.decode import generate_2
        return generate_2(self, inputs, max_length=max_length)
