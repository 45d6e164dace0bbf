"""LossManager -- drop-in for footprints/training/losses.py:14-92, computed by ONE fused HIP kernel.

`LossManager(depth_range, prior)(predictions, targets) -> dict` with the reference's 21 keys
(('visible_ground'|'all_ground'|'depth'|'ground_depth'|'loss', scale) x 4 scales + 'loss').  The kernel
produces the 21 scalars and d(loss)/d(pred) for all 16 prediction planes in a single pass (the reference runs
~100 elementwise kernels + 16 reductions).  Only losses['loss'] carries gradient -- it is the only one the
trainer differentiates (training/train.py:153-155); the per-scale entries are detached monitoring values.

Like the reference (losses.py:90) the call MUTATES `predictions`, adding the 20 tuple-keyed visualisation
tensors; they are plumbing for the tensorboard logger and are computed lazily (plain torch elementwise ops on
the device) only when `compute_viz` is True.
"""
import torch

from .. import ops
from ..utils import sigmoid_to_depth

SCALES = ("1/8", "1/4", "1/2", "1/1")
LOSS_KEYS = [(n, s) for s in SCALES for n in ("visible_ground", "all_ground", "depth", "ground_depth", "loss")] + ["loss"]
TARGET_KEYS = ("visible_ground", "all_ground", "depth", "ground_depth", "moving_object_mask", "depth_mask")


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mgr, targets, p8, p4, p2, p1):
        preds = [p.contiguous() for p in (p8, p4, p2, p1)]
        need = any(p.requires_grad for p in (p8, p4, p2, p1))
        out = torch.empty(21, device=p1.device)
        dp = [torch.empty_like(p) for p in preds] if need else None
        ops.loss_fwd_bwd(preds, targets, out, dp, (mgr.min_depth, mgr.max_depth), mgr.footprint_prior_weight)
        ctx.dp = dp
        return out

    @staticmethod
    def backward(ctx, gout):
        if ctx.dp is None:
            return (None,) * 6
        s = gout[20]                       # only the total loss is differentiable
        return (None, None) + tuple(d * s for d in ctx.dp)


class LossManager:
    def __init__(self, depth_range, footprint_prior_weight, compute_viz=True):
        self.min_depth, self.max_depth = depth_range
        self.footprint_prior_weight = footprint_prior_weight
        self.compute_viz = compute_viz

    def __call__(self, predictions, targets):
        for k in SCALES:
            if not predictions[k].is_cuda:
                raise RuntimeError("footprints_amd.LossManager has no CPU path (oracle/ holds the CPU restatement for tests)")
        tg = {k: targets[k].contiguous().float() for k in TARGET_KEYS}
        vec = _LossFn.apply(self, tg, *[predictions[k] for k in SCALES])
        losses = {}
        for i, key in enumerate(LOSS_KEYS):
            losses[key] = vec[i] if key == "loss" else vec[i].detach()
        if self.compute_viz:
            predictions.update(self._viz(predictions))          # losses.py:90
        return losses

    def _viz(self, predictions):
        out = {}
        with torch.no_grad():
            for k in SCALES:
                o = predictions[k]
                out[("visible_ground", k)] = torch.sigmoid(o[:, 0])
                out[("all_ground", k)] = torch.sigmoid(o[:, 1])
                out[("depth", k)] = sigmoid_to_depth(o[:, 2], self.min_depth, self.max_depth)
                gd = sigmoid_to_depth(o[:, 3], self.min_depth, self.max_depth)
                out[("ground_depth", k)] = gd
                out[("ground_depth_masked", k)] = gd * (out[("all_ground", k)] > 0.5).float()   # losses.py:76-78
        return out
