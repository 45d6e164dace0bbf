"""Synthetic per-sample inputs of the data-path fixture (G10): shared by the generator (which writes them as the jpg-free files the
reference's KITTIDataset reads) and by the tests (which feed the same arrays to the oracle / the HIP kernels)."""
import numpy as np

from oracle import filler

N_SAMPLES, H, W, SEED = 12, 64, 96, 1234


def sample_inputs(i):
    """-> (uint8 [H,W,3] image, dict of float64 [H,W] maps as they are BEFORE the flip)"""
    img = (filler.uniform("g10:%d:image" % i, (H, W, 3)) * 256).astype(np.uint8)
    if i % 4 == 1:
        img[:, : W // 3] = img[:, :1]                              # flat / grey regions
        img[: H // 4, :, 1] = img[: H // 4, :, 0]
        img[: H // 4, :, 2] = img[: H // 4, :, 0]
    maps = {
        "visible_ground": filler.uniform("g10:%d:vg" % i, (H, W)).astype(np.float64),
        "ground_depth": (filler.uniform("g10:%d:gd" % i, (H, W), 0.0, 30.0) * filler.bernoulli("g10:%d:gdv" % i, (H, W), 0.5)).astype(np.float64),
        "depth_mask": filler.bernoulli("g10:%d:dm" % i, (H, W), 0.02).astype(np.float64),
        "disparity": (filler.uniform("g10:%d:disp" % i, (H, W), 0.0, 60.0) * filler.bernoulli("g10:%d:dv" % i, (H, W), 0.9)).astype(np.float64),
        "moving_objects": filler.bernoulli("g10:%d:mov" % i, (H, W), 0.05).astype(np.float64),
    }
    maps["disparity"][0, :5] = 1.25                                # disp - 1.25 == 0: the `disp - (disp == 0)` branch of utils.py:31
    return img, maps


def sample_inputs_matterport(i):
    """Matterport flavour: 'depth_raw' = the 16-bit depth png's values; ground depth with exact 0.1 entries (missing pixels,
    matterport_dataset.py:73) and values beyond the 10 m cut (:76)"""
    img, _ = sample_inputs(i)
    gd = (filler.uniform("g11:%d:gd" % i, (H, W), 0.0, 14.0) * filler.bernoulli("g11:%d:gdv" % i, (H, W), 0.6)).astype(np.float64)
    gd[1, :7] = 0.1
    maps = {
        "visible_ground": filler.uniform("g11:%d:vg" % i, (H, W)).astype(np.float64),
        "ground_depth": gd,
        "depth_mask": filler.bernoulli("g11:%d:dm" % i, (H, W), 0.02).astype(np.float64),
        "depth_raw": np.floor(filler.uniform("g11:%d:depth" % i, (H, W), 0.0, 40000.0)).astype(np.float64),
    }
    return img, maps
