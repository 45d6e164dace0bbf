"""GPU: the device-side data path (csrc/data_path.hip) against (a) the G10 fixture -- the reference's own
KITTIDataset.__getitem__ outputs -- bit for bit, (b) the oracle (oracle/data_path.py, itself pinned to the real Pillow) on larger
random batches incl. every op order, (c) the double-buffered loader against per-batch assembly."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params_from(flip, jit):
    from footprints_amd.datasets import AugParams
    p = AugParams()
    p.flip = int(flip)
    if jit is not None:
        order, factors = jit
        p.n_ops = 4
        for k in range(4):
            p.ops[k] = order[k]
            p.factor[k] = factors[k]
        p.hue_shift = int(factors[3] * 255) & 0xFF
    return p


def test_g10_device_path_equals_reference_dataset_bit_for_bit():
    from footprints_amd.datasets import DeviceBatchAssembler, draw_augmentation
    from tests.golden.data_inputs import N_SAMPLES, SEED, H, W, sample_inputs
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g10_data_path.npz"))
    rng = random.Random(SEED)
    from oracle import data_path as D
    D.jitter_params(rng)                                          # the dataset constructor's probe of get_params (footprint_dataset.py:41-42)
    samples = [sample_inputs(i) for i in range(N_SAMPLES)]
    params = [draw_augmentation(True, rng) for _ in range(N_SAMPLES)]     # same RNG stream as the reference -> same decisions
    asm = DeviceBatchAssembler(N_SAMPLES, H, W, dataset="kitti", map_dtype=np.float64)
    batch = asm.collect(asm.submit(samples, params))
    torch.cuda.synchronize()
    for i in range(N_SAMPLES):
        assert (params[i].flip, int(params[i].n_ops > 0)) == tuple(int(v) for v in g["flags"][i])
        for k, v in batch.items():
            ref = g["%d.%s" % (i, k)]
            got = v[i].cpu().numpy()
            assert np.array_equal(ref, got), (i, k, float(np.abs(ref - got).max()))


def test_g11_device_path_equals_reference_matterport_dataset_bit_for_bit():
    from footprints_amd.datasets import DeviceBatchAssembler, draw_augmentation
    from oracle import data_path as D
    from tests.golden.data_inputs import N_SAMPLES, SEED, H, W, sample_inputs_matterport
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g11_data_path_matterport.npz"))
    rng = random.Random(SEED)
    D.jitter_params(rng)
    samples = [sample_inputs_matterport(i) for i in range(N_SAMPLES)]
    params = [draw_augmentation(True, rng) for _ in range(N_SAMPLES)]
    asm = DeviceBatchAssembler(N_SAMPLES, H, W, dataset="matterport", map_dtype=np.float64)
    batch = asm.collect(asm.submit(samples, params))
    torch.cuda.synchronize()
    for i in range(N_SAMPLES):
        for k, v in batch.items():
            assert np.array_equal(g["%d.%s" % (i, k)], v[i].cpu().numpy()), (i, k)


@pytest.mark.parametrize("B,H,W", [(3, 37, 53), (12, 192, 640)])
def test_device_path_equals_oracle_every_op_order(B, H, W):
    import itertools
    from footprints_amd.datasets import DeviceBatchAssembler
    from oracle import data_path as D
    nrng, rng = np.random.default_rng(5), random.Random(5)
    orders = list(itertools.permutations(range(4)))
    for rep in range(2 if H > 100 else 8):
        samples, params, refs = [], [], []
        for b in range(B):
            img = nrng.integers(0, 256, (H, W, 3), dtype=np.uint8)
            if b % 3 == 0:
                img[:, : W // 2] = img[:, :1]
            maps = {"visible_ground": nrng.random((H, W)), "ground_depth": nrng.random((H, W)) * 30 * (nrng.random((H, W)) < 0.5),
                    "depth_mask": (nrng.random((H, W)) < 0.05).astype(np.float64), "disparity": nrng.random((H, W)) * 60 * (nrng.random((H, W)) < 0.9),
                    "moving_objects": (nrng.random((H, W)) < 0.05).astype(np.float64)}
            maps["disparity"][0, :3] = 1.25
            flip = rng.random() > 0.5
            jit = None
            if rng.random() > 0.3:
                _, factors = D.jitter_params(rng)
                jit = (list(orders[(rep * B + b) % 24]), factors)
            samples.append((img, maps))
            params.append(_params_from(flip, jit))
            refs.append(D.assemble_kitti(img, maps, flip, jit, W))
        asm = DeviceBatchAssembler(B, H, W, dataset="kitti", map_dtype=np.float64)
        batch = asm.collect(asm.submit(samples, params))
        torch.cuda.synchronize()
        for b in range(B):
            for k, v in batch.items():
                assert np.array_equal(refs[b][k], v[b].cpu().numpy()), (rep, b, k)
        # float32 label inputs: same algebra on inputs rounded once -> float32-level agreement, masks identical away from threshold ties
        asm32 = DeviceBatchAssembler(B, H, W, dataset="kitti", map_dtype=np.float32)
        b32 = asm32.collect(asm32.submit(samples, params))
        torch.cuda.synchronize()
        assert torch.equal(b32["image"], batch["image"])
        assert torch.allclose(b32["ground_depth"], batch["ground_depth"], rtol=1e-6, atol=1e-6)
        # depth = f b / (disp - 1.25) amplifies the float32 rounding of disp by 1 / |disp - 1.25|: compare where that is <= 1
        for b in range(B):
            d = samples[b][1]["disparity"]
            d = d[:, ::-1] if params[b].flip else d
            well = torch.from_numpy(np.ascontiguousarray(np.abs(d - 1.25) >= 1.0)).cuda()
            assert torch.allclose(b32["depth"][b][well], batch["depth"][b][well], rtol=1e-5, atol=1e-6), b


def test_device_loader_double_buffering_matches_single_batches():
    from footprints_amd.datasets import DeviceBatchAssembler, DeviceLoader, draw_augmentation
    B, H, W, NB = 2, 32, 64, 5
    nrng = np.random.default_rng(11)
    batches = []
    for _ in range(NB):
        ss = []
        for _ in range(B):
            img = nrng.integers(0, 256, (H, W, 3), dtype=np.uint8)
            maps = {k: nrng.random((H, W)) for k in ("visible_ground", "ground_depth", "depth_mask", "disparity", "moving_objects")}
            ss.append((img, maps))
        batches.append(ss)
    asm = DeviceBatchAssembler(B, H, W, dataset="kitti")
    got = []
    for batch in DeviceLoader(batches, asm, is_train=True, rng=random.Random(9)):
        x = {k: v.clone() for k, v in batch.items()}           # "the step": read the slot
        got.append(x)
    rng = random.Random(9)
    one = DeviceBatchAssembler(B, H, W, dataset="kitti", slots=1)
    assert len(got) == NB
    for ss, g in zip(batches, got):
        params = [draw_augmentation(True, rng) for _ in ss]
        ref = one.collect(one.submit(ss, params))
        torch.cuda.synchronize()
        for k in ref:
            assert torch.equal(ref[k], g[k]), k
        one.release(0)


def test_matterport_label_algebra_against_oracle():
    from footprints_amd.datasets import DeviceBatchAssembler
    from oracle import data_path as D
    B, H, W = 2, 32, 48
    nrng = np.random.default_rng(2)
    samples, params, refs = [], [], []
    for b in range(B):
        img = nrng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        gd = nrng.random((H, W)) * 14
        gd[0, :4] = 0.1
        maps = {"visible_ground": nrng.random((H, W)), "ground_depth": gd, "depth_mask": (nrng.random((H, W)) < 0.05).astype(np.float64),
                "depth_raw": nrng.integers(0, 40000, (H, W)).astype(np.float64)}
        samples.append((img, maps))
        params.append(_params_from(b == 1, None))
        refs.append(D.assemble_matterport(img, maps, b == 1, None))
    asm = DeviceBatchAssembler(B, H, W, dataset="matterport")
    batch = asm.collect(asm.submit(samples, params))
    torch.cuda.synchronize()
    for b in range(B):
        for k, v in batch.items():
            assert np.array_equal(refs[b][k], v[b].cpu().numpy()), (b, k)
