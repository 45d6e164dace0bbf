"""Loss of the ground-segmentation trainer on the device -- what footprints/preprocessing/segmentation/train.py:184-193 (bilinear up-sizing
of the four logit maps) and segmentation/evaluation.py:39-58 (per-image masked BCE-with-logits means, averaged over the scales, batch
mean for backprop, per-key running lists) compute with a dozen torch ops, as ONE differentiable call on `Segmentor`'s outputs:

    seg_loss = SegmentationLoss()
    loss = seg_loss(model(image), ground_mask, loss_mask)     # scalar, autograd-capable (the gradient is computed in the same launches)
    loss.backward()
    seg_loss.tracked()                                        # {'ground_loss_0' .. 'ground_loss_3', 'loss'} -> means since the last call

HIP kernels: csrc/seg_loss.hip (fp_seg_loss_fwd_bwd).  No CPU path."""
from collections import defaultdict

import torch

from ... import ops


class _SegLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ground_mask, loss_mask, holder, *preds):
        B = ground_mask.shape[0]
        losses = torch.empty(5 * B + 1, device=ground_mask.device)
        grads = [torch.empty_like(p, memory_format=torch.contiguous_format) for p in preds]
        maps = [p if (p.stride(3) == 1 and p.stride(2) == p.shape[3]) else p.contiguous() for p in preds]
        if any(m.stride(0) != g.stride(0) for m, g in zip(maps, grads)):      # channel slices of wider buffers: dense copies for the kernel
            maps = [m.contiguous() for m in maps]
        ops.seg_loss_fwd_bwd(maps, ground_mask.contiguous().float(), loss_mask.contiguous().float(), losses, dpreds=grads)
        ctx.grads = grads
        holder.append(losses)
        return losses[5 * B].clone()

    @staticmethod
    def backward(ctx, gout):
        return (None, None, None) + tuple(g * gout for g in ctx.grads)


class SegmentationLoss:
    """callable with the reference Evaluator's bookkeeping (segmentation/evaluation.py:13-58): per-key lists of per-image losses,
    `tracked()` = their means since the last call (the reference's get_tracked_losses)"""

    def __init__(self):
        self._lists = defaultdict(list)

    def __call__(self, outputs, ground_mask, loss_mask):
        if len(outputs) != 4:
            raise ValueError("SegmentationLoss expects the four logit maps of Segmentor.forward")
        holder = []
        loss = _SegLossFn.apply(ground_mask, loss_mask, holder, *outputs)
        B = ground_mask.shape[0]
        per = holder[0].detach()
        for s in range(4):
            self._lists["ground_loss_%d" % s].append(per[s * B:(s + 1) * B])
        self._lists["loss"].append(per[4 * B:5 * B])
        return loss

    def tracked(self):
        out = {k: torch.cat(v).mean() for k, v in self._lists.items()}
        self._lists = defaultdict(list)
        return out
