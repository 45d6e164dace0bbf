"""CPU restatement of the per-sample data path between the dataset readers and the network.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Restates, with the reference file:line each piece follows:
  * FootprintsDataset.preprocess            footprints/datasets/footprint_dataset.py:55-65   (colour jitter, ToTensor, float maps, all_ground)
  * flips                                   footprint_dataset.py:73-75,84-85
  * KITTIDataset.__getitem__ label algebra  footprints/datasets/kitti_dataset.py:66-67,72-73,86-97,105-112
  * MatterportDataset label algebra         footprints/datasets/matterport_dataset.py:69-78,93-97

THIRD-PARTY arithmetic (absent from /root/reference, restated from the published algorithms -- "parity unpinned by the reference",
pinned instead against the libraries present in this image):
  * torchvision==0.4.2 (environment.yml:9) `transforms.ColorJitter` / `functional.adjust_*` / `ToTensor`: restated below on top of
    the REAL Pillow (`jitter_pil`), call order and RNG draws as in torchvision/transforms/transforms.py (get_params: four
    random.uniform draws in the order brightness, contrast, saturation, hue, then random.shuffle of the four closures);
  * Pillow (environment.yml pins 6.2.1; this image has 12.2): `ImageEnhance.{Brightness,Contrast,Color}` = `Image.blend` with a
    black / mean-grey / per-pixel-grey image, `convert("L")`, `convert("HSV")` and back.  `jitter_np` is the byte-exact numpy
    restatement of that integer / float arithmetic, verified against Pillow over ALL 2^24 colours and all byte pairs
    (tests/test_oracle_data_path.py); the HIP kernel is compared with it bit for bit.
"""
import random

import numpy as np

BRIGHTNESS, CONTRAST, SATURATION, HUE = 0, 1, 2, 3
JITTER_RANGES = ((0.8, 1.2), (0.8, 1.2), (0.8, 1.2), (-0.1, 0.1))      # footprint_dataset.py:37-40


# ----------------------------------------------------------------------------------------------------------------------
# torchvision 0.4.2 ColorJitter (third-party restatement)
# ----------------------------------------------------------------------------------------------------------------------
def jitter_params(rng=random):
    """ColorJitter.get_params: -> (order, factors): `order` = the four op ids in application order, factors indexed by op id.
    Consumes the RNG exactly like torchvision 0.4.2: uniform x4 (brightness, contrast, saturation, hue), then shuffle."""
    factors = [rng.uniform(lo, hi) for lo, hi in JITTER_RANGES]
    order = [BRIGHTNESS, CONTRAST, SATURATION, HUE]
    rng.shuffle(order)
    return order, factors


def sample_augmentation(is_train, rng=random):
    """the draws of KITTIDataset.__getitem__ (kitti_dataset.py:55-56) followed by ColorJitter's own (only when color_aug fires,
    footprint_dataset.py:58-59 -> transforms.ColorJitter.__call__)"""
    do_flip = bool(is_train and rng.random() > 0.5)
    color_aug = bool(is_train and rng.random() > 0.5)
    return do_flip, color_aug


def hue_shift_byte(hue_factor):
    """np.uint8(hue_factor * 255) of torchvision's adjust_hue: truncation toward zero, then wrap-around modulo 256"""
    return int(hue_factor * 255) & 0xFF


def jitter_pil(img, order, factors):
    """apply the jitter with the real Pillow (what torchvision's functional.adjust_* do)"""
    from PIL import Image, ImageEnhance
    for op in order:
        f = factors[op]
        if op == BRIGHTNESS:
            img = ImageEnhance.Brightness(img).enhance(f)
        elif op == CONTRAST:
            img = ImageEnhance.Contrast(img).enhance(f)
        elif op == SATURATION:
            img = ImageEnhance.Color(img).enhance(f)
        else:
            h, s, v = img.convert("HSV").split()
            np_h = np.array(h, dtype=np.uint8)
            np_h = (np_h.astype(np.int32) + hue_shift_byte(f)).astype(np.uint8)       # uint8 wrap-around
            img = Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB")
    return img


# ----------------------------------------------------------------------------------------------------------------------
# Pillow's byte arithmetic, in numpy
# ----------------------------------------------------------------------------------------------------------------------
def blend_np(in1, in2, alpha):
    """Image.blend(im1, im2, alpha) per byte (libImaging/Blend.c): float32 `in1 + alpha * (in2 - in1)`, truncated; clipped to
    [0, 255] first when alpha is outside [0, 1]"""
    al = np.float32(alpha)
    t = in1.astype(np.int32).astype(np.float32) + al * (in2.astype(np.int32) - in1.astype(np.int32)).astype(np.float32)
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32))).astype(np.uint8)


def luma_np(rgb):
    """convert("L") (libImaging/Convert.c L24): (R*19595 + G*38470 + B*7471 + 0x8000) >> 16"""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def rgb2hsv_np(rgb):
    """convert("HSV") (libImaging/Convert.c rgb2hsv_row): float variables, double literals"""
    f32, f64 = np.float32, np.float64
    r, g, b = (rgb[..., i].astype(np.int32) for i in range(3))
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    with np.errstate(all="ignore"):
        cr = (maxc - minc).astype(f32)
        s = cr / maxc.astype(f32)
        rc, gc, bc = ((maxc - c).astype(f32) / cr for c in (r, g, b))
        h = np.where(r == maxc, bc - gc,
                     np.where(g == maxc, ((f64(2.0) + rc.astype(f64)) - bc.astype(f64)).astype(f32),
                              ((f64(4.0) + gc.astype(f64)) - rc.astype(f64)).astype(f32)))
        h = np.fmod(h.astype(f64) / 6.0 + 1.0, 1.0).astype(f32)
        uh = np.clip((h.astype(f64) * 255.0).astype(np.int64), 0, 255)
        us = np.clip((s.astype(f64) * 255.0).astype(np.int64), 0, 255)
    grey = minc == maxc
    return np.stack([np.where(grey, 0, uh), np.where(grey, 0, us), maxc], -1).astype(np.uint8)


def hsv2rgb_np(hsv):
    """HSV -> RGB (libImaging/Convert.c hsv2rgb): float f / fs, double products, C round()"""
    f32, f64 = np.float32, np.float64
    h, s, v = hsv[..., 0].astype(f32), hsv[..., 1].astype(np.int32), hsv[..., 2].astype(np.int32)
    hd = h.astype(f64) * 6.0 / 255.0
    i = np.floor(hd).astype(np.int32)
    f = (hd - i.astype(f32).astype(f64)).astype(f32).astype(f64)
    fs = (s.astype(f32).astype(f64) / 255.0).astype(f32).astype(f64)
    vf = v.astype(f32).astype(f64)

    def cround(x):
        return np.where(x >= 0, np.floor(x + 0.5), -np.floor(-x + 0.5)).astype(np.int64)
    up, uq, ut = (np.clip(cround(vf * t), 0, 255) for t in (1.0 - fs, 1.0 - fs * f, 1.0 - fs * (1.0 - f)))
    k = i % 6
    r = np.choose(k, [v, uq, up, up, ut, v])
    g = np.choose(k, [ut, v, v, uq, up, up])
    b = np.choose(k, [up, up, ut, v, v, uq])
    grey = s == 0
    return np.stack([np.where(grey, v, r), np.where(grey, v, g), np.where(grey, v, b)], -1).astype(np.uint8)


def jitter_np(rgb, order, factors):
    """the whole jitter on a uint8 [H,W,3] array, byte-exact with jitter_pil"""
    img = rgb
    for op in order:
        f = factors[op]
        if op == BRIGHTNESS:
            img = blend_np(np.zeros_like(img), img, f)
        elif op == CONTRAST:
            lum = luma_np(img).astype(np.int64)
            mean = int(lum.sum() / lum.size + 0.5)                                   # int(ImageStat.Stat(L).mean[0] + 0.5)
            img = blend_np(np.full_like(img, mean), img, f)
        elif op == SATURATION:
            img = blend_np(np.repeat(luma_np(img)[..., None], 3, -1), img, f)
        else:
            hsv = rgb2hsv_np(img)
            hsv[..., 0] = (hsv[..., 0].astype(np.int32) + hue_shift_byte(f)).astype(np.uint8)
            img = hsv2rgb_np(hsv)
    return img


# ----------------------------------------------------------------------------------------------------------------------
# the assembled sample (reference schema, datasets/footprint_dataset.py:55-65)
# ----------------------------------------------------------------------------------------------------------------------
def assemble_kitti(image_u8, maps, do_flip, jitter, width, no_depth_mask=False, project_down_baseline=False, moving_objects="ours",
                   footprint_threshold=0.75, baseline=0.54):
    """image_u8 [H,W,3] uint8 (already resized, not flipped); maps: float64 [H,W] arrays as they leave load_and_resize_npy WITHOUT
    the flip: 'visible_ground' (probability), 'ground_depth', 'depth_mask' (after filter_depth_mask), 'disparity' (resized and
    rescaled, before the -1.25), 'moving_objects'.  jitter = None or (order, factors).  -> dict of float32 arrays (reference schema).
    Every step cites kitti_dataset.py; arithmetic in float64 like numpy there, cast to float32 at the end (torch.tensor(val).float())."""
    img = image_u8[:, ::-1] if do_flip else image_u8                                  # footprint_dataset.py:73-75
    m = {k: (v[:, ::-1] if do_flip else v).astype(np.float64) for k, v in maps.items()}   # footprint_dataset.py:84-85
    if jitter is not None:
        img = jitter_np(np.ascontiguousarray(img), *jitter)                            # footprint_dataset.py:58-59
    vg = m["visible_ground"] > footprint_threshold                                    # kitti_dataset.py:66-67
    gd = np.ones_like(m["ground_depth"]) if project_down_baseline else m["ground_depth"].copy()   # :72-73
    dm = m["depth_mask"] * (0 if no_depth_mask else 1)                                # :86-87
    gd[dm.astype(bool)] = 0                                                           # :90
    disp = m["disparity"] - 1.25                                                      # :94-96
    fx = 0.58 * width                                                                 # :23-27 (K[0,0] * width)
    with np.errstate(divide="ignore"):
        depth = np.float32(fx) * baseline / (disp - (disp == 0))                      # utils.py:31 (self.K is float32)
    depth[depth < 0] = 0                                                              # utils.py:32
    mov = m["moving_objects"] if moving_objects == "ours" else np.zeros_like(gd)      # :99-103
    mov = mov * (1 - vg)                                                              # :106
    mov = mov * (1 - dm)                                                              # :108
    out = {"image": (np.ascontiguousarray(img).transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)),   # ToTensor
           "visible_ground": vg.astype(np.float32), "depth": depth.astype(np.float32), "ground_depth": gd.astype(np.float32),
           "moving_object_mask": mov.astype(np.float32), "depth_mask": dm.astype(np.float32)}
    out["all_ground"] = ((out["ground_depth"] + out["visible_ground"]) > 0).astype(np.float32)   # footprint_dataset.py:64
    return out


def assemble_matterport(image_u8, maps, do_flip, jitter, no_depth_mask=False, footprint_threshold=0.75, depth_scaling=0.25e-3):
    """matterport_dataset.py:40-110 on resized arrays: 'visible_ground', 'ground_depth', 'depth_mask', 'depth_raw' (16-bit PNG values)"""
    img = image_u8[:, ::-1] if do_flip else image_u8
    m = {k: (v[:, ::-1] if do_flip else v).astype(np.float64) for k, v in maps.items()}
    if jitter is not None:
        img = jitter_np(np.ascontiguousarray(img), *jitter)
    vg = m["visible_ground"] > footprint_threshold                                    # matterport_dataset.py:60
    depth = m["depth_raw"] * depth_scaling                                            # :70
    gd = m["ground_depth"].copy()
    gd[gd == 0.1] = 0                                                                 # :73
    gd *= (gd < 10.0)                                                                 # :76
    mov = np.zeros_like(depth)                                                        # :79
    dm = m["depth_mask"] * (0 if no_depth_mask else 1)                                # :93-94
    gd[dm.astype(bool)] = 0                                                           # :97
    out = {"image": (np.ascontiguousarray(img).transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)),
           "visible_ground": vg.astype(np.float32), "depth": depth.astype(np.float32), "ground_depth": gd.astype(np.float32),
           "moving_object_mask": mov.astype(np.float32), "depth_mask": dm.astype(np.float32)}
    out["all_ground"] = ((out["ground_depth"] + out["visible_ground"]) > 0).astype(np.float32)
    return out
