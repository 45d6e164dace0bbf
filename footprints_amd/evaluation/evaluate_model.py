"""Test-set metrics -- counterpart of the reference's footprints/evaluation/evaluate_model.py.

The reference scores one file at a time with numpy on the host.  Here a batch of predictions is scored by two device
reductions (`fp_eval_mask_counts`: integer confusion counts, `fp_eval_depth_sums`: float64 error sums, one workgroup per image)
and the host turns them into the same per-image dictionaries (`iou / precision / recall / f1`, `a1 / abs_rel / sq_rel / rmse`,
nan where the reference returns nan) and the same nan-means.  Thresholds and dtypes follow the reference: float16 predictions
(the inference pass's file format) are thresholded and converted to depth in float16 like numpy does.
Ground-truth download and the Matterport convex-hull pre-processing stay outside this build.
"""
import os

import numpy as np
import torch

from .. import ops

VISIBLE_GROUND, HIDDEN_GROUND, DEPTH, HIDDEN_DEPTH = 0, 1, 2, 3      # channels of the prediction arrays (evaluate_model.py:17-21)
nan = float("nan")


def _dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a)) if not torch.is_tensor(a) else a
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda(non_blocking=True).contiguous()


def _mask_dict(c):
    """evaluate_model.py:80-99 from the confusion counts (n_true, tp, fp, fn)"""
    n_true, tp, fp, fn = (int(v) for v in c)
    if n_true == 0:
        return {key: nan for key in ["iou", "precision", "recall", "f1"]}
    union = n_true + fp
    iou = tp / union if union > 0 else 0
    precision = tp / (tp + fp) if (tp + fp) > 0 else 0
    recall = tp / (tp + fn) if (tp + fn) > 0 else 0
    f1 = 2 * (precision * recall) / (precision + recall) if (precision + recall) > 0 else 0
    return {"iou": iou, "precision": precision, "recall": recall, "f1": f1}


def mask_scores(preds, ground_truths, free_spaces):
    """preds [B,4,H,W] or [B,H,W] (float16 / float32 sigmoid outputs), ground_truths [B,H,W] (bool or float),
    free_spaces [B,H,W] bool -> list of {"freespace": {...}, "footprint": {...}} like evaluate()'s all_scores (:165-168)."""
    p = _dev(preds)
    if p.dim() == 4:
        p = p[:, HIDDEN_GROUND]
    gt = _dev(ground_truths, torch.float32)
    fs = _dev(free_spaces, torch.uint8)
    a = ops.eval_mask_counts(p, gt).cpu().numpy()
    b = ops.eval_mask_counts(p, gt, region=fs, invert=True).cpu().numpy()
    return [{"freespace": _mask_dict(a[i]), "footprint": _mask_dict(b[i])} for i in range(p.shape[0])]


def depth_scores(preds, ground_truths, max_depth=20):
    """preds [B,4,H,W] or [B,H,W] sigmoid disparities, ground_truths [B,H,W] hidden depths (0 = none) -> list of
    {"a1", "abs_rel", "sq_rel", "rmse"} (evaluate_model.py:170-175, 50-69)."""
    p = _dev(preds)
    if p.dim() == 4:
        p = p[:, HIDDEN_DEPTH]
    s = ops.eval_depth_sums(p, _dev(ground_truths, torch.float32), clip=(0.5, float(max_depth))).cpu().numpy()
    out = []
    for n, n_a1, sq, abs_rel, sq_rel in s:
        if n == 0:
            out.append({key: nan for key in ["a1", "abs_rel", "sq_rel", "rmse"]})
        else:
            out.append({"a1": n_a1 / n, "abs_rel": abs_rel / n, "sq_rel": sq_rel / n, "rmse": float(np.sqrt(sq / n))})
    return out


def summarise(all_scores, metric):
    """the numbers evaluate() prints (evaluate_model.py:179-194)"""
    if metric == "iou":
        return {"freespace_iou": float(np.nanmean([s["freespace"]["iou"] for s in all_scores])),
                "freespace_f1": float(np.nanmean([s["freespace"]["f1"] for s in all_scores])),
                "footprint_iou": float(np.nanmean([s["footprint"]["iou"] for s in all_scores])),
                "footprint_f1": float(np.nanmean([s["footprint"]["f1"] for s in all_scores]))}
    if metric == "depth":
        return {k: float(np.nanmean([s[k] for s in all_scores])) for k in ("a1", "rmse", "abs_rel", "sq_rel")}
    raise Exception("unknown metric {}".format(metric))


def evaluate_arrays(preds, ground_truths, free_spaces, metric, batch=64):
    """Score in-memory arrays in batches; returns (all_scores, summary)."""
    all_scores = []
    for i in range(0, len(preds), batch):
        sl = slice(i, i + batch)
        if metric == "iou":
            all_scores += mask_scores(preds[sl], ground_truths[sl], free_spaces[sl])
        elif metric == "depth":
            all_scores += depth_scores(preds[sl], ground_truths[sl])
        else:
            raise Exception("unknown metric {}".format(metric))
    return all_scores, summarise(all_scores, metric)


def _load_mask(filepath):
    from PIL import Image
    if not os.path.exists(filepath):
        raise FileNotFoundError(filepath)
    return np.asarray(Image.open(filepath).convert("L")) > 128          # evaluate_model.py:41-46


def evaluate(pred_folder, datatype, metric, ground_truth_dir, split_file="splits/matterport/test.txt", batch=32):
    """Folder evaluation with the reference's file layout (evaluate_model.py:131-194); ground truths must already be on disk."""
    if datatype == "kitti":
        if metric == "depth":
            raise ValueError("The kitti annotations do not contain depth data for evaluation")
        names = list(range(697))
    elif datatype == "matterport":
        with open(split_file, "r") as fh:
            names = [xx.split() for xx in fh.read().splitlines()]
    else:
        raise Exception("unknown datatype {}".format(datatype))
    all_scores = []
    for i in range(0, len(names), batch):
        P, G, F = [], [], []
        for name in names[i:i + batch]:
            if datatype == "kitti":
                d = os.path.join(ground_truth_dir, "kitti_ground_truth", "kitti_ground_truth")
                G.append(_load_mask(os.path.join(d, "{:05d}_combined.png".format(name))))
                F.append(_load_mask(os.path.join(d, "{:05d}_ground.png".format(name))))
                P.append(np.load(os.path.join(pred_folder, "{:03d}.npy".format(name))))
            else:
                d = os.path.join(ground_truth_dir, "matterport_ground_truth", "matterport_ground_truth")
                G.append(np.load(os.path.join(d, "{}_{}_{}_{}_groundtruth.npy".format(*name))))
                F.append(np.load(os.path.join(d, "{}_{}_{}_{}_freespace.npy".format(*name))) > 0.5)
                P.append(np.load(os.path.join(pred_folder, "{}".format(name[0]), "{}_{}_{}.npy".format(*name[1:]))))
        scores, _ = evaluate_arrays(np.stack(P), np.stack(G), np.stack(F), metric, batch=batch)
        all_scores += scores
    return summarise(all_scores, metric)
