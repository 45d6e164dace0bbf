"""Evaluator of the ground-segmentation trainer -- drop-in for footprints/preprocessing/segmentation/evaluation.py:13-58.

Same interface (`compute_losses(predictions, ground_mask, loss_mask)` with predictions keyed `(type, scale)`, `get_tracked_losses()`),
same arithmetic: per-image masked BCE-with-logits means, averaged over the four scales, batch mean returned for backprop.  These are a
handful of elementwise torch ops on 1-channel maps (host plumbing around the network, like the reference); the network itself is
`footprints_amd.preprocessing.segmentation.network.Segmentor` on the HIP engine."""
from collections import defaultdict

import torch
from torch import nn


class Evaluator:
    def __init__(self):
        self.loss_func = nn.BCEWithLogitsLoss(reduction="none")
        self.tracked_loss = defaultdict(list)

    def _reset_losses(self):
        self.tracked_loss = defaultdict(list)

    def get_tracked_losses(self):
        losses = self.tracked_loss.copy()
        self._reset_losses()
        for key, val in losses.items():
            losses[key] = torch.cat(val, dim=0).mean()
        return losses

    def compute_losses(self, predictions, ground_mask, loss_mask):
        total_loss = 0
        for prediction_type, scale in predictions.keys():
            pred = predictions[(prediction_type, scale)]
            loss = self.loss_func(pred, ground_mask)
            valid_pix = loss_mask.sum(dim=[1, 2])
            loss = (loss * loss_mask).sum(dim=[1, 2]) / (valid_pix + 1e-7)
            self.tracked_loss["{}_loss_{}".format(prediction_type, scale)].append(loss.detach())
            total_loss = total_loss + loss
        total_loss = total_loss / 4
        self.tracked_loss["loss"].append(total_loss.detach())
        return total_loss.mean()


def upsize_predictions(outputs, height, width):
    """segmentation/train.py:184-190: the four logit maps bilinearly up-sized (align_corners=False) to the input resolution, keyed
    ('ground', scale) with the channel dimension removed"""
    predictions = {}
    for scale, out in enumerate(outputs):
        out = nn.functional.interpolate(out, size=(height, width), mode="bilinear", align_corners=False)
        predictions[("ground", scale)] = out.squeeze(1)
    return predictions
