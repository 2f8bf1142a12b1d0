"""`tasks.mt3_net_segmem.MT3NetSegMem` — drop-in for tasks/mt3_net_segmem.py:12-68 (memory prepended to the
decoder input, models/t5_segmem.py)."""
from models.t5_segmem import T5SegMem
from tasks.mt3_base import SegMemTask


class MT3NetSegMem(SegMemTask):
    MODEL = T5SegMem
