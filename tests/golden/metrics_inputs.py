"""Inputs of the G7 metrics fixture, regenerated from the counter-hash filler on both sides (generator and tests)."""
import numpy as np

from oracle import filler

N, H, W = 6, 48, 64
MASK_KEYS = ("iou", "precision", "recall", "f1")
DEPTH_KEYS = ("a1", "abs_rel", "sq_rel", "rmse")


def metrics_inputs(pred_dtype):
    """-> pred [N,4,H,W] (sigmoid outputs, pred_dtype), ground truths and free-space masks in both dataset flavours"""
    pred = filler.uniform("g7.pred", (N, 4, H, W)).astype(np.float32)
    # threshold edge cases on the hidden-ground channel: exactly 0.5 and its float16 / float32 neighbours
    edge = np.array([0.5, 0.5 - 2.0 ** -12, 0.5 + 2.0 ** -11, 0.5 - 2.0 ** -25, 0.5 + 2.0 ** -24, 0.49987793, 0.50024414], np.float32)
    pred[:, 1, 0, :edge.size] = edge
    pred[:, 1, 1, :edge.size] = edge
    pred[:, 3, 0, :4] = np.array([0.0, 1.0, 2.0 ** -14, 0.999], np.float32)
    pred = pred.astype(pred_dtype)
    gt_kitti = filler.bernoulli("g7.gt.kitti", (N, H, W), 0.3) > 0.5                      # bool, like load_mask()
    gt_kitti[4] = False                                                                   # no hidden ground: nan scores
    free = filler.bernoulli("g7.free", (N, H, W), 0.6) > 0.5
    free[3, :, :] = True
    gt_mp = filler.bernoulli("g7.gt.mp", (N, H, W), 0.4) * filler.uniform("g7.gt.mp.v", (N, H, W), 0.0, 1.2)   # float hidden ground
    gt_mp[:, 2, :6] = np.array([0.1, 0.10000001, 0.9, 0.89999998, 0.05, 1.0], np.float32)
    gt_mp = gt_mp.astype(np.float32)
    gt_depth = (filler.bernoulli("g7.gt.depth.m", (N, H, W), 0.5) * filler.uniform("g7.gt.depth", (N, H, W), 0.0, 30.0)).astype(np.float32)
    gt_depth[5] = 0.0                                                                     # nothing to evaluate: nan scores
    return pred, gt_kitti, gt_mp, free, gt_depth
