"""`tasks.mt3_base.MT3Base` — drop-in for tasks/mt3_base.py:8-49.

Derives from `pytorch_lightning.LightningModule` when Lightning is importable (the reference's
trainer then drives these classes unchanged); otherwise from a minimal stand-in with the hooks the
MI355X trainer (`mrmt3.trainer`) calls, so the task surface works without Lightning installed.
"""
import glob

import torch.nn as nn

try:  # optional dependency
    import pytorch_lightning as pl
    from pytorch_lightning.utilities.rank_zero import rank_zero_only
    _Base = pl.LightningModule
except ImportError:  # Lightning absent (as in this image)
    pl = None

    def rank_zero_only(fn):
        def wrapped(self, *a, **kw):
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
                return None
            return fn(self, *a, **kw)
        return wrapped

    class _Base(nn.Module):
        """The slice of LightningModule the tasks use: `log`, `current_epoch`, `global_step`."""

        def __init__(self):
            super().__init__()
            self.logged = {}
            self.current_epoch = 0
            self.global_step = 0

        def log(self, name, value, **kwargs):
            self.logged[name] = value

        @classmethod
        def load_from_checkpoint(cls, path, **kwargs):
            from mrmt3.checkpoint import read_checkpoint
            obj = cls(**kwargs)
            obj.model.load_state_dict(read_checkpoint(path)["state_dict"], strict=False)
            return obj


class MT3Base(_Base):
    """Base class for MT3 related experiments"""

    def __init__(self, config, optim_cfg, eval_cfg=None):
        super().__init__()
        self.config = config
        self.optim_cfg = optim_cfg
        self.eval_cfg = eval_cfg

    def forward(self, *args, **kwargs):
        raise NotImplementedError

    def training_step(self, batch, batch_idx):
        raise NotImplementedError

    def validation_step(self, batch, batch_idx):
        raise NotImplementedError

    @rank_zero_only
    def on_validation_epoch_end(self):
        # Rank-0 transcription F1 needs the reference's test.py/evaluate.py stack (mir_eval,
        # note_seq, real Slakh audio): out of scope for the hot path.
        # The hook is kept so Lightning's loop finds it; it evaluates only when that stack exists.
        if self.eval_cfg is None:
            return
        try:
            from test import get_scores  # the reference's evaluation entry, if present on sys.path
        except Exception:
            return
        if self.current_epoch >= self.eval_cfg.eval_after_num_epoch and \
                self.current_epoch % self.eval_cfg.eval_per_epoch == 0:
            eval_audio_dir = sorted(glob.glob(self.eval_cfg.audio_dir))
            if self.eval_cfg.eval_first_n_examples:
                eval_audio_dir = eval_audio_dir[:self.eval_cfg.eval_first_n_examples]
            self.model.eval()
            scores = get_scores(model=self.model, eval_audio_dir=eval_audio_dir, eval_dataset="Slakh",
                                ground_truth_midi_dir=self.eval_cfg.midi_dir, verbose=False)
            self.log('val_f1_flat', scores['Onset F1'], on_step=False, on_epoch=True, prog_bar=True)
            self.log('val_f1_midi_class', scores['Onset + program F1 (midi_class)'], on_step=False, on_epoch=True)
            self.log('val_f1_full', scores['Onset + program F1 (full)'], on_step=False, on_epoch=True)

    def configure_optimizers(self):
        raise NotImplementedError

    # ---- shared by the concrete tasks ------------------------------------------------------------------
    def _cfg(self, name, default=None):
        c = self.config
        return c[name] if isinstance(c, dict) else getattr(c, name, default)

    def _opt(self, name):
        c = self.optim_cfg
        return c[name] if isinstance(c, dict) else getattr(c, name)

    def _cosine_optimizers(self):
        from torch.optim import AdamW
        from utils import get_cosine_schedule_with_warmup
        optimizer = AdamW(self.model.parameters(), self._opt("lr"))
        warmup_step = int(self._opt("warmup_steps"))
        print('warmup step: ', warmup_step)
        schedule = {
            'scheduler': get_cosine_schedule_with_warmup(
                optimizer=optimizer, num_warmup_steps=warmup_step,
                num_training_steps=self._opt("num_steps_per_epoch") * self._opt("num_epochs"),
                min_lr=self._opt("min_lr")),
            'interval': 'step',
            'frequency': 1,
        }
        return [optimizer], [schedule]


class SegMemTask(MT3Base):
    """What the three segment-memory tasks share (the reference spells it out once per file,
    tasks/mt3_net_segmem*.py): the model is built from `MODEL` with the two memory hyper-parameters of the
    model config, a batch is `(inputs, targets)` plus `targets_prev` when `WITH_PREV`, the loss is the plain
    token cross-entropy and the optimiser the cosine-with-warm-up AdamW."""
    MODEL = None
    WITH_PREV = False

    def __init__(self, config, optim_cfg, eval_cfg=None):
        super().__init__(config, optim_cfg, eval_cfg=eval_cfg)
        self.model = self.MODEL(config=self.config, segmem_num_layers=self._cfg("segmem_num_layers", 1),
                                segmem_length=self._cfg("segmem_length", 64))

    def forward(self, *args, **kwargs):
        return self.model.forward(*args, **kwargs)

    def _loss(self, batch):
        from tasks.mt3_net import _ce
        names = ("inputs", "labels", "targets_prev") if self.WITH_PREV else ("inputs", "labels")
        fields = dict(zip(names, batch))
        return _ce(self.forward(**fields), fields["labels"])

    def training_step(self, batch, batch_idx):
        loss = self._loss(batch)
        self.log("train_loss", loss, prog_bar=True, on_step=True, on_epoch=False, sync_dist=True)
        return loss

    def validation_step(self, batch, batch_idx):
        import torch
        with torch.no_grad():
            self.log("val_loss", self._loss(batch), prog_bar=True, on_step=False, on_epoch=True, sync_dist=True)

    def configure_optimizers(self):
        return self._cosine_optimizers()
