"""Evaluator -- drop-in for footprints/training/evaluation.py:14-67.

Same interface and bookkeeping; the one change is that the 21 loss scalars of a step leave the device in ONE
copy (they are 21 views of one vector) instead of the reference's 21 blocking `.cpu()` calls per step
(evaluation.py:38-43).
"""
from collections import defaultdict

import torch

from .losses import LossManager


class Evaluator:
    def __init__(self, depth_range, footprint_prior, compute_viz=True):
        self.accumulated_train_losses = defaultdict(list)
        self.accumulated_val_losses = defaultdict(list)
        self.loss_manager = LossManager(depth_range, footprint_prior, compute_viz=compute_viz)

    def compute_losses(self, inputs, outputs, mode="train", return_batch_loss=False):
        losses = self.loss_manager(predictions=outputs, targets=inputs)
        keys = list(losses.keys())
        host = torch.stack([losses[k].detach() for k in keys]).cpu()       # one D2H copy
        acc = self.accumulated_train_losses if mode == "train" else self.accumulated_val_losses if mode == "val" else None
        if acc is not None:
            for i, k in enumerate(keys):
                acc[k].append(host[i])
        if return_batch_loss:
            return losses

    def get_averaged_losses(self, mode, reset=True):
        averaged = {}
        attr = "accumulated_train_losses" if mode == "train" else "accumulated_val_losses"
        for k, v in getattr(self, attr).items():
            averaged[k] = float(torch.stack(v).mean().numpy())
        if reset:
            setattr(self, attr, defaultdict(list))
        return averaged
