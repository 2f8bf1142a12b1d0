"""`tasks.mt3_net_segmem.MT3NetSegMem` — drop-in for tasks/mt3_net_segmem.py:12-68."""
import torch
import torch.nn as nn

from models.t5_segmem import T5SegMem
from tasks.mt3_base import MT3Base
from tasks.mt3_net import _ce


class MT3NetSegMem(MT3Base):
    def __init__(self, config, optim_cfg, eval_cfg=None):
        super().__init__(config, optim_cfg, eval_cfg=eval_cfg)
        self.model: nn.Module = T5SegMem(
            config=self.config,
            segmem_num_layers=self._cfg("segmem_num_layers", 1),
            segmem_length=self._cfg("segmem_length", 64),
        )

    def forward(self, *args, **kwargs):
        return self.model.forward(*args, **kwargs)

    def training_step(self, batch, batch_idx):
        inputs, targets = batch
        loss = _ce(self.forward(inputs=inputs, labels=targets), targets)
        self.log('train_loss', loss, prog_bar=True, on_step=True, on_epoch=False, sync_dist=True)
        return loss

    @torch.no_grad()
    def validation_step(self, batch, batch_idx):
        inputs, targets = batch
        loss = _ce(self.forward(inputs=inputs, labels=targets), targets)
        self.log('val_loss', loss, prog_bar=True, on_step=False, on_epoch=True, sync_dist=True)

    def configure_optimizers(self):
        return self._cosine_optimizers()
