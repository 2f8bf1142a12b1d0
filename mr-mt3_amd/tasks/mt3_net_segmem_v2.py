"""`tasks.mt3_net_segmem_v2.MT3NetSegMemV2` — drop-in for tasks/mt3_net_segmem_v2.py:12-64 (memory appended to the
encoder states, previous tokens derived from the labels, models/t5_segmem_v2.py)."""
from models.t5_segmem_v2 import T5SegMemV2
from tasks.mt3_base import SegMemTask


class MT3NetSegMemV2(SegMemTask):
    MODEL = T5SegMemV2
