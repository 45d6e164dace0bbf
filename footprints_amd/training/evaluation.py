"""Evaluator -- drop-in for footprints/training/evaluation.py:14-67.

Same interface and bookkeeping (`compute_losses(inputs, outputs, mode, return_batch_loss)`, `get_averaged_losses(mode, reset)`,
the `accumulated_{train,val}_losses` attributes).  Two changes, both about host/device traffic:
  * the 21 loss scalars of a step are 21 views of ONE device vector (the fused loss kernel writes them together), so they are
    kept as that vector and leave the device in one copy when an average is asked for -- the reference does 21 blocking
    `.cpu()` calls per step (evaluation.py:38-43), i.e. 21 pipeline drains per training step;
  * `add_loss_vector(vec, mode)` lets the fused `TrainStep` (which never builds the dict) feed the same accumulators.
Averages are computed exactly like the reference's: per key, `torch.stack(list of fp32 scalars).mean()` on the host.
"""
from collections import defaultdict

import torch

from .losses import LOSS_KEYS, LossManager


class Evaluator:
    def __init__(self, depth_range, footprint_prior, compute_viz=True):
        self.accumulated_train_losses = defaultdict(list)
        self.accumulated_val_losses = defaultdict(list)
        self._pending = {"train": [], "val": []}          # device-side 21-vectors not yet copied to the host
        self.loss_manager = LossManager(depth_range, footprint_prior, compute_viz=compute_viz)

    def add_loss_vector(self, vec, mode="train"):
        """vec: the 21 losses of one batch in LOSS_KEYS order (device tensor; cloned, no synchronisation)"""
        if mode in self._pending:
            self._pending[mode].append(vec.detach().clone())

    def compute_losses(self, inputs, outputs, mode="train", return_batch_loss=False):
        losses = self.loss_manager(predictions=outputs, targets=inputs)
        if mode in self._pending:
            self._pending[mode].append(torch.stack([losses[k].detach() for k in LOSS_KEYS]))
        if return_batch_loss:
            return losses

    def _flush(self, mode):
        pend = self._pending[mode]
        if pend:
            host = torch.stack(pend).cpu()                 # ONE D2H copy for every batch since the last average
            acc = self.accumulated_train_losses if mode == "train" else self.accumulated_val_losses
            for row in host:
                for i, k in enumerate(LOSS_KEYS):
                    acc[k].append(row[i].clone())
            self._pending[mode] = []

    def get_averaged_losses(self, mode, reset=True):
        averaged = {}
        if mode not in self._pending:
            return averaged
        self._flush(mode)
        attr = "accumulated_train_losses" if mode == "train" else "accumulated_val_losses"
        for k, v in getattr(self, attr).items():
            averaged[k] = float(torch.stack(v).mean().numpy())      # evaluation.py:54-55
        if reset:
            setattr(self, attr, defaultdict(list))
        return averaged
