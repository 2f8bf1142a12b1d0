"""`tasks.mt3_net_segmem_v2_with_prev_finetune.MT3NetSegMemV2WithPrevFineTune` — drop-in for
tasks/mt3_net_segmem_v2_with_prev_finetune.py:11-20: same model, plain AdamW without a schedule."""
from torch.optim import AdamW

from tasks.mt3_net_segmem_v2_with_prev import MT3NetSegMemV2WithPrev


class MT3NetSegMemV2WithPrevFineTune(MT3NetSegMemV2WithPrev):
    def configure_optimizers(self):
        return AdamW(self.model.parameters(), self._opt("lr"))
