"""`models.t5_segmem_v2` — drop-in for the reference's V2 segment-memory model
(models/t5_segmem_v2.py:38-233): memory of the previous batch row concatenated to the encoder
output and reached through cross-attention."""
import torch

from mrmt3.module import MT3Module


class T5SegMemV2(MT3Module):
    VARIANT = "segmem_v2"

    def __init__(self, config, segmem_num_layers: int = 1, segmem_length: int = 64, compute_dtype=None):
        super().__init__(config, segmem_num_layers=segmem_num_layers, segmem_length=segmem_length,
                         compute_dtype=compute_dtype or torch.bfloat16)

    def generate_songs(self, songs, max_length=1024, **kwargs):
        """Several recordings at once, one decode-batch row per recording (each keeps its own memory chain);
        row results equal `generate` on that recording alone.  Not in the reference, which is sequential."""
        from mrmt3.decode import generate_songs
        return generate_songs(self, songs, max_length=max_length)
