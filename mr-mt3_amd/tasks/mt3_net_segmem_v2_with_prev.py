"""`tasks.mt3_net_segmem_v2_with_prev.MT3NetSegMemV2WithPrev` — drop-in for
tasks/mt3_net_segmem_v2_with_prev.py:12-72, the MR-MT3 task: batches are `(inputs, targets, targets_prev)`."""
from models.t5_segmem_v2_with_prev import T5SegMemV2WithPrev
from tasks.mt3_base import SegMemTask


class MT3NetSegMemV2WithPrev(SegMemTask):
    MODEL = T5SegMemV2WithPrev
    WITH_PREV = True
