"""`tasks.mt3_net` — drop-in for tasks/mt3_net.py: MT3Net (:12-68) and MT3NetWeightedLoss (:75-165)."""
import torch
import torch.nn as nn

from models.t5 import T5ForConditionalGeneration
from tasks.mt3_base import MT3Base


def _ce(lm_logits, targets):
    # tasks/mt3_net.py:32-35 — torch ops on the kernel-produced logits; autograd reaches the HIP
    # backward through mrmt3.module._ModelFn.
    return nn.CrossEntropyLoss(ignore_index=-100)(lm_logits.view(-1, lm_logits.size(-1)), targets.view(-1))


class MT3Net(MT3Base):
    def __init__(self, config, optim_cfg, eval_cfg=None):
        super().__init__(config, optim_cfg, eval_cfg=eval_cfg)
        self.model: nn.Module = T5ForConditionalGeneration(self.config)

    def forward(self, *args, **kwargs):
        return self.model.forward(*args, **kwargs)

    def training_step(self, batch, batch_idx):
        inputs, targets = batch
        lm_logits = self.forward(inputs=inputs, labels=targets)
        loss = _ce(lm_logits, targets)
        self.log('train_loss', loss, prog_bar=True, on_step=True, on_epoch=False, sync_dist=True)
        return loss

    @torch.no_grad()
    def validation_step(self, batch, batch_idx):
        inputs, targets = batch
        lm_logits = self.forward(inputs=inputs, labels=targets)
        loss = _ce(lm_logits, targets)
        self.log('val_loss', loss, prog_bar=True, on_step=False, on_epoch=True, sync_dist=True)

    def configure_optimizers(self):
        return self._cosine_optimizers()


class MT3NetWeightedLoss(MT3Base):
    def __init__(self, config, optim_cfg, eval_cfg=None):
        super().__init__(config, optim_cfg, eval_cfg=eval_cfg)
        self.model: nn.Module = T5ForConditionalGeneration(self.config)

    def forward(self, *args, **kwargs):
        return self.model.forward(*args, **kwargs)

    def _losses(self, lm_logits, targets):
        # tasks/mt3_net.py:96-108: program tokens (1135..1262) are added a second time with weight 2
        flat = targets.view(-1)
        inst = (flat >= 1135) & (flat <= 1262)
        nonpad = flat != -100
        raw = nn.CrossEntropyLoss(reduction="none")(lm_logits.view(-1, lm_logits.size(-1)), flat)
        li, lm = torch.masked_select(raw, inst), torch.masked_select(raw, nonpad)
        loss = (lm.sum() + 2 * li.sum()) / (li.shape[0] + lm.shape[0])
        return loss, lm.sum() / lm.shape[0], li.sum() / li.shape[0]

    def training_step(self, batch, batch_idx):
        inputs, targets = batch
        loss, other, inst = self._losses(self.forward(inputs=inputs, labels=targets), targets)
        self.log('train_loss_other', other, prog_bar=True, on_step=True, on_epoch=False, sync_dist=True)
        self.log('train_loss_inst', inst, prog_bar=True, on_step=True, on_epoch=False, sync_dist=True)
        self.log('train_loss', loss, prog_bar=True, on_step=True, on_epoch=False, sync_dist=True)
        return loss

    @torch.no_grad()
    def validation_step(self, batch, batch_idx):
        inputs, targets = batch
        loss, other, inst = self._losses(self.forward(inputs=inputs, labels=targets), targets)
        self.log('val_loss_other', other, prog_bar=True, on_step=False, on_epoch=True, sync_dist=True)
        self.log('val_loss_inst', inst, prog_bar=True, on_step=False, on_epoch=True, sync_dist=True)
        self.log('val_loss', loss, prog_bar=True, on_step=False, on_epoch=True, sync_dist=True)

    def configure_optimizers(self):
        return self._cosine_optimizers()
