"""`tasks.mt3_net_segmem_v2_with_prev.MT3NetSegMemV2WithPrev` — drop-in for
tasks/mt3_net_segmem_v2_with_prev.py:12-72 (batches are `(inputs, targets, targets_prev)`)."""
import torch
import torch.nn as nn

from models.t5_segmem_v2_with_prev import T5SegMemV2WithPrev
from tasks.mt3_base import MT3Base
from tasks.mt3_net import _ce


class MT3NetSegMemV2WithPrev(MT3Base):
    def __init__(self, config, optim_cfg, eval_cfg=None):
        super().__init__(config, optim_cfg, eval_cfg=eval_cfg)
        self.model: nn.Module = T5SegMemV2WithPrev(
            config=self.config,
            segmem_num_layers=self._cfg("segmem_num_layers", 1),
            segmem_length=self._cfg("segmem_length", 64),
        )

    def forward(self, *args, **kwargs):
        return self.model.forward(*args, **kwargs)

    def training_step(self, batch, batch_idx):
        inputs, targets, targets_prev = batch
        loss = _ce(self.forward(inputs=inputs, labels=targets, targets_prev=targets_prev), targets)
        self.log('train_loss', loss, prog_bar=True, on_step=True, on_epoch=False, sync_dist=True)
        return loss

    @torch.no_grad()
    def validation_step(self, batch, batch_idx):
        inputs, targets, targets_prev = batch
        loss = _ce(self.forward(inputs=inputs, labels=targets, targets_prev=targets_prev), targets)
        self.log('val_loss', loss, prog_bar=True, on_step=False, on_epoch=True, sync_dist=True)

    def configure_optimizers(self):
        return self._cosine_optimizers()
