"""Flip + colour jitter + ToTensor + label algebra on the GPU, fed by ONE pinned, double-buffered H2D copy per batch.

Reference (per sample, on DataLoader workers): footprints/datasets/footprint_dataset.py:55-65 (`preprocess`: ColorJitter, ToTensor,
float maps, `all_ground`), :73-75 / :84-85 (flips), kitti_dataset.py:55-56 (augmentation draws), :66-112 (label algebra),
matterport_dataset.py:69-97.  At ~650 img/s per GPU the reference's 8 PIL workers cannot feed one MI355X, let alone eight; here the
host only hands over what the file readers produce -- the resized uint8 image and the resized label maps of every sample -- plus a
36-byte parameter record per sample, and two kernels (csrc/data_path.hip) assemble the whole batch in the reference's schema.

Random decisions stay on the host and consume Python's `random` exactly like the reference does (flip draw, colour-aug draw, then
torchvision 0.4.2's ColorJitter.get_params: four uniforms in the order brightness, contrast, saturation, hue and one shuffle), so a
seeded run makes the same decisions as the reference pipeline.  Byte arithmetic is bit-exact with Pillow 12 (tests).
"""
import ctypes as C
import random

import numpy as np
import torch

from .. import _lib, ops

BRIGHTNESS, CONTRAST, SATURATION, HUE = 0, 1, 2, 3
JITTER_RANGES = ((0.8, 1.2), (0.8, 1.2), (0.8, 1.2), (-0.1, 0.1))          # footprint_dataset.py:37-40
MAP_KEYS = {"kitti": ("visible_ground", "ground_depth", "depth_mask", "disparity", "moving_objects"),
            "matterport": ("visible_ground", "ground_depth", "depth_mask", "depth_raw")}
OUT_KEYS = ("visible_ground", "depth", "ground_depth", "moving_object_mask", "depth_mask", "all_ground")


class AugParams(C.Structure):
    """fp_aug_params (include/footprints_hip.h)"""
    _fields_ = [("flip", C.c_int32), ("n_ops", C.c_int32), ("ops", C.c_int32 * 4), ("factor", C.c_float * 4), ("hue_shift", C.c_int32),
                ("pad", C.c_int32)]


def draw_augmentation(is_train=True, rng=random):
    """the reference's draws for one sample, in its order (kitti_dataset.py:55-56, then ColorJitter.get_params only when the colour
    augmentation fires) -> AugParams"""
    p = AugParams()
    p.flip = int(bool(is_train and rng.random() > 0.5))
    color_aug = bool(is_train and rng.random() > 0.5)
    if color_aug:
        factors = [rng.uniform(lo, hi) for lo, hi in JITTER_RANGES]
        order = [BRIGHTNESS, CONTRAST, SATURATION, HUE]
        rng.shuffle(order)
        p.n_ops = 4
        for k in range(4):
            p.ops[k] = order[k]
            p.factor[k] = factors[k]
        p.hue_shift = int(factors[HUE] * 255) & 0xFF                     # np.uint8(hue_factor * 255): truncation, wrap-around
    return p


class DeviceBatchAssembler:
    """Pinned staging buffers (one set per slot), a copy stream, and the two assembly kernels.  Three slots by default: while the
    network works on batch i (slot a) and batch i + 1 sits assembled in slot b, the host can already fill slot c with batch i + 2 --
    with two slots the refill of a slot has to wait for the step that last read it, and the GPU idles for the host's memcpy.

        asm = DeviceBatchAssembler(12, 192, 640, dataset="kitti")
        slot = asm.submit(samples, params)        # host memcpy into pinned memory + async H2D + kernels on the copy stream
        batch = asm.collect(slot)                 # the consumer's stream waits for that slot's event; dict with the reference keys

    samples: list of (image uint8 [H,W,3], {map name: [H,W] array}) as the file readers deliver them (resized, NOT flipped; the depth
    mask already through filter_depth_mask).  map_dtype float64 reproduces the reference's numpy arithmetic bit for bit; float32
    halves the H2D bytes (inputs rounded once before the same float64 algebra)."""

    def __init__(self, batch_size, height, width, dataset="kitti", map_dtype=np.float64, slots=3, no_depth_mask=False,
                 project_down_baseline=False, moving_objects_method="ours", footprint_threshold=0.75, baseline=0.54,
                 depth_scaling=0.25e-3, device="cuda", stream=None):
        if dataset not in MAP_KEYS:
            raise ValueError("dataset must be 'kitti' or 'matterport'")
        _lib.load()
        assert _lib.load().fp_aug_params_bytes() == C.sizeof(AugParams)
        self.B, self.H, self.W, self.dataset = batch_size, height, width, dataset
        self.map_dtype = np.dtype(map_dtype)
        if self.map_dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError("map_dtype must be float32 or float64")
        self.keys = MAP_KEYS[dataset]
        self.no_depth_mask, self.pdb = bool(no_depth_mask), bool(project_down_baseline)
        self.use_moving = dataset == "kitti" and moving_objects_method == "ours"
        self.threshold = float(footprint_threshold)
        # focal * baseline exactly as the reference evaluates it (utils.py:31 with self.K[0, 0], a float32 scalar, kitti_dataset.py:23-28):
        # left to numpy so that the scalar promotion rules of the installed numpy apply, as they do to the reference's own code
        self.fxb = float(np.float32(0.58 * width) * baseline) if dataset == "kitti" else 0.0
        self.depth_scaling = float(depth_scaling)
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            # "cuda" means the CURRENT device at construction, pinned to an explicit index: under a launcher the current device
            # changes when DistContext.from_env selects LOCAL_RANK, and a bare "cuda" stream / buffer created before that lives on GPU 0
            self.device = torch.device("cuda", torch.cuda.current_device())
        # the copy / assembly stream.  A process owns four hardware queues and the engine uses all four (footprints_amd/engine.py): a
        # stream of its own is a fifth one and shares a queue with whichever engine stream the runtime picks -- a 3.6 ms H2D copy then
        # sits in front of that stream's kernels (637 vs 851 img/s on two boxes of the pool).  Pass the engine's decoder weight-gradient
        # stream (`model.engine().dwg[0]`): the next batch is staged before the current step is launched, the copy is done long before
        # that stream's first kernel of the step (6 ms in), and nothing on the critical path ever waits behind it.
        self.stream = stream if stream is not None else torch.cuda.Stream(device=self.device)
        npx = batch_size * height * width
        tdt = torch.float64 if self.map_dtype == np.float64 else torch.float32
        self.slots = []
        for _ in range(slots):
            h_img = torch.empty((batch_size, height, width, 3), dtype=torch.uint8).pin_memory()
            h_maps = torch.empty((len(self.keys), batch_size, height, width), dtype=tdt).pin_memory()
            h_par = torch.empty(batch_size * C.sizeof(AugParams), dtype=torch.uint8).pin_memory()
            self.slots.append(dict(
                h_img=h_img, h_maps=h_maps, h_par=h_par,
                d_img=torch.empty_like(h_img, device=self.device), d_maps=torch.empty_like(h_maps, device=self.device),
                d_par=torch.empty_like(h_par, device=self.device), sums=torch.zeros(batch_size, dtype=torch.int64, device=self.device),
                image=torch.empty((batch_size, 3, height, width), device=self.device),
                out=torch.empty((len(OUT_KEYS), batch_size, height, width), device=self.device),
                ready=torch.cuda.Event(), consumed=None, launched=False))
        self._next = 0
        assert npx > 0

    def fill(self, samples, params):
        """host half of staging one batch: copy the samples and augmentation draws into the next slot's pinned buffers (60 MB of numpy
        copies for a KITTI batch with float64 maps -- DeviceLoader runs this in a worker thread).  Waits only for the slot's previous
        H2D copies.  Returns the slot index for launch()."""
        if len(samples) != self.B or len(params) != self.B:
            raise ValueError("expected %d samples" % self.B)
        i = self._next
        self._next = (self._next + 1) % len(self.slots)
        s = self.slots[i]
        if s["launched"]:
            s["ready"].synchronize()                         # the pinned buffers are free once the slot's last copies have landed
        img_np, maps_np = s["h_img"].numpy(), s["h_maps"].numpy()
        for b, (img, maps) in enumerate(samples):
            img_np[b] = img
            for k, key in enumerate(self.keys):
                maps_np[k, b] = maps[key]
        arr = (AugParams * self.B)(*params)
        s["h_par"].numpy()[:] = np.frombuffer(bytes(arr), dtype=np.uint8)
        return i

    def launch(self, i):
        """device half: H2D copies + assembly kernels of a filled slot on the copy stream, behind the consumer's last read of the slot's
        device buffers (an event wait on the stream: the host does not block)."""
        s = self.slots[i]
        if s["consumed"] is not None:
            self.stream.wait_event(s["consumed"])            # the consumer finished reading this slot's outputs
        with ops.on_stream(self.stream):
            s["d_img"].copy_(s["h_img"], non_blocking=True)
            s["d_maps"].copy_(s["h_maps"], non_blocking=True)
            s["d_par"].copy_(s["h_par"], non_blocking=True)
            lib = _lib.load()
            st = ops.stream()
            _lib.check(lib.fp_assemble_images(s["d_img"].data_ptr(), s["d_par"].data_ptr(), s["sums"].data_ptr(), s["image"].data_ptr(),
                                              self.B, self.H, self.W, st), "fp_assemble_images")
            m, o = s["d_maps"], s["out"]
            kitti = self.dataset == "kitti"
            _lib.check(lib.fp_assemble_labels(m[0].data_ptr(), m[1].data_ptr(), m[2].data_ptr(), m[3].data_ptr(),
                                              m[4].data_ptr() if kitti else None, int(self.map_dtype == np.float64), s["d_par"].data_ptr(),
                                              o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), o[4].data_ptr(), o[5].data_ptr(),
                                              self.B, self.H, self.W, 0 if kitti else 1, int(self.no_depth_mask), int(self.pdb),
                                              int(self.use_moving), self.threshold, self.fxb, self.depth_scaling, st), "fp_assemble_labels")
            s["ready"].record(self.stream)
        s["launched"] = True
        return i

    def submit(self, samples, params):
        """stage one batch (fill + launch); returns the slot index to pass to collect()"""
        return self.launch(self.fill(samples, params))

    def collect(self, slot):
        """batch dict (reference schema) of the slot; the current stream waits for the slot's kernels.  The tensors are views of the
        slot's buffers: they stay valid until the slot is submitted again (call `release(slot)` after the step that used them was
        queued, or let DeviceLoader do it)."""
        s = self.slots[slot]
        torch.cuda.current_stream().wait_event(s["ready"])
        batch = {"image": s["image"]}
        for k, key in enumerate(OUT_KEYS):
            batch[key] = s["out"][k]
        return batch

    def release(self, slot):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.slots[slot]["consumed"] = ev


class DeviceLoader:
    """Iterable of device batches over a host sample source: while the network works on batch i, batch i + 1 is being copied and
    assembled on the copy stream (the assembler's second slot).

    source: iterable yielding lists of `batch_size` samples (image uint8 [H,W,3], maps dict); rng: Python RNG for the reference's
    augmentation draws (the module-level `random` by default, like the reference)."""

    def __init__(self, source, assembler, is_train=True, rng=random):
        self.source, self.asm, self.is_train, self.rng = source, assembler, is_train, rng

    def __len__(self):
        return len(self.source)

    def shard(self, rank, world):
        """per-rank view for parallel.ShardedLoader: the SOURCE is sharded by index (a rank decodes, stages and assembles only its own
        batches), the assembler and its slots stay this process's"""
        if not hasattr(self.source, "shard"):
            raise TypeError("DeviceLoader.shard: the sample source has no shard(rank, world)")
        return DeviceLoader(self.source.shard(rank, world), self.asm, self.is_train, self.rng)

    def __iter__(self):
        # one worker thread does the host half of staging (augmentation draws in batch order + the copies into pinned memory: numpy
        # releases the GIL for them), one batch ahead of the batch whose copies and kernels are in flight, two ahead of the consumer
        from concurrent.futures import ThreadPoolExecutor
        it = iter(self.source)

        def fill_next():
            try:
                samples = next(it)
            except StopIteration:
                return None
            return self.asm.fill(samples, [draw_augmentation(self.is_train, self.rng) for _ in samples])

        with ThreadPoolExecutor(1) as ex:
            pending = fill_next()
            if pending is None:
                return
            self.asm.launch(pending)
            fut = ex.submit(fill_next)
            while True:
                batch = self.asm.collect(pending)
                nxt = fut.result()
                if nxt is not None:
                    self.asm.launch(nxt)                  # the OTHER slot: copied and assembled while the consumer works on `pending`
                    fut = ex.submit(fill_next)
                yield batch
                self.asm.release(pending)                 # the step that used it has been queued: its completion frees the slot
                if nxt is None:
                    return
                pending = nxt


class SyntheticSampleSource:
    """`steps` batches of host samples with the shapes / dtypes the file readers deliver (stand-in for the KITTI reader, which is out
    of scope): a small pool of random samples, cycled."""

    def __init__(self, batch_size, height, width, steps, seed=10, pool=24):
        rng = np.random.default_rng(seed)
        self.B, self.steps, self.first, self.stride = batch_size, steps, 0, 1
        self.pool = []
        for _ in range(pool):
            maps = {"visible_ground": rng.random((height, width)), "ground_depth": rng.random((height, width)) * 30 * (rng.random((height, width)) < 0.5),
                    "depth_mask": (rng.random((height, width)) < 0.1).astype(np.float64), "disparity": rng.random((height, width)) * 60,
                    "moving_objects": (rng.random((height, width)) < 0.05).astype(np.float64)}
            self.pool.append((rng.integers(0, 256, (height, width, 3), dtype=np.uint8), maps))
        self.dataset = range(steps * batch_size)

    def __len__(self):
        return len(range(self.first, self.steps, self.stride))

    def shard(self, rank, world):
        import copy
        v = copy.copy(self)
        v.steps = (self.steps // world) * world
        v.first, v.stride = rank, world
        return v

    def __iter__(self):
        for i in range(self.first, self.steps, self.stride):
            yield [self.pool[(i * self.B + j) % len(self.pool)] for j in range(self.B)]
