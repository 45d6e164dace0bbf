"""The training hot loop -- drop-in counterpart of footprints/training/train.py:145-227 (`run_epoch` /
`process_batch`), without the dataset / tensorboard plumbing that is out of scope.

`TrainStep` is the MI355X fast path of `process_batch` + `zero_grad` + `backward` + `optimiser.step`
(train.py:150-156): it drives the HIP engine's explicit forward / loss / backward / Adam schedule on static
buffers (no autograd graph, no per-step allocation, one D2H copy of the 21 losses only when asked).
`TrainManager` keeps the reference's loop structure (step counter, 100-step console line, epoch-end
checkpoint + StepLR) around it for synthetic or user-provided loaders.
"""
import os
import time

import torch

from .. import ops
from ..model_manager import ModelManager
from ..parallel import DistContext, GradReducer, broadcast_state
from .evaluation import Evaluator
from .losses import LOSS_KEYS, SCALES, TARGET_KEYS

SEED = 10   # train.py:33


_GRAPH = bool(int(os.environ.get("FP_GRAPH", "0")))     # hipGraph replay of the whole step (single-GPU TrainStep); opt-in, see below
# recorded launch plan (csrc/plan.cpp): after two eager steps the step's ~740 launches + ~150 stream-ordering edges are recorded once
# per distinct set of input buffers and replayed from C with their original streams; FP_PLAN=0 keeps issuing every step from Python
_PLAN = bool(int(os.environ.get("FP_PLAN", "1")))
# Adam in pieces (FP_ADAM_STAGED=1, opt-in): a stage's slice of the flat buffers is updated on a side stream as soon as that stage's gradients
# are complete, under the rest of the backward pass, instead of one launch alone on the GPU after it (element-wise: the same bits,
# tests/test_gpu_switches.py).  Single-GPU steps with the concurrent schedule only -- in data-parallel steps the update has to follow the
# buckets' all-reduces.  Measured -0.03 ms of 11.37 per step (profiles/round4_notes.md section 18): the 141 us launch leaves the end of the step,
# but the kernels it now runs beside give most of that back; not the default until a timeline says where the rest went.
_ADAM_STAGED = bool(int(os.environ.get("FP_ADAM_STAGED", "0")))
_ADAM_STAGE_MIN = 1 << 20        # stages below this many elements stay in the closing launch


class TrainStep:
    """One full training step (forward + fused loss + backward + Adam) on static buffers, no autograd graph.

    graph=True (or FP_GRAPH=1; single-GPU only): after two eager steps the whole step -- ~770 launches on five streams -- is
    captured once into a hipGraph and replayed; the batch is copied into static input buffers and the seven Adam scalars of the
    step are uploaded before each replay, so results are bit-identical to eager (tested).  It is OFF by default: measured on
    ROCm 7.2 / MI355X the replay costs the host as much as the eager launches (16.4 vs 14.9 ms per step), does not shorten the
    dependent-launch gaps (one stream: 497 img/s both ways) and runs the forked streams with less overlap (509 vs 578 img/s)."""

    def __init__(self, model, optimiser, depth_range=(0.1, 100.0), footprint_prior=0.25, distributed=False, graph=None, plan=None):
        self.model, self.optimiser = model, optimiser
        self.use_graph = (_GRAPH and not distributed) if graph is None else bool(graph)
        self.use_plan = (_PLAN if plan is None else bool(plan)) and not self.use_graph
        self._plans = {}            # (input pointers, shapes) -> (plan handle, stage marks, node index where Adam starts, node count)
        self._plan_shape = None
        self._plan_gen = None       # ops.alloc_generation() the plans were recorded at
        self._graph = None
        self._eager_steps = 0
        self._static = None
        self._hyper = None
        self.depth_range, self.prior = depth_range, footprint_prior
        self.eng = model.engine()
        self.losses = torch.zeros(21, device=self.eng.device)
        self.outputs = None
        self.dpreds = None
        self.reducer = None
        self._adam_ranges = None    # stage -> (lo, hi) of the flat buffers, for the staged update
        if _ADAM_STAGED and not distributed and not self.use_graph:
            from ..parallel import bucket_ranges
            eng = self.eng
            self._adam_ranges = {s: (lo, hi) for s, lo, hi in bucket_ranges(eng.live_names, eng.offsets, eng.flat_grad.numel(),
                                                                            max_elems=eng.flat_grad.numel())}
        if distributed:
            from ..parallel import _COMM_STREAM
            eng = self.eng
            # the bucket all-reduces run on one of the engine's own streams (default: the mask decoder's weight-gradient stream, idle
            # once the decoders are done) -- a stream of their own would be a fifth one and share a hardware queue with a busy one
            comm_stream = {"aux": eng.aux, "wg": eng.wg, "dwg0": eng.dwg[0], "dwg1": eng.dwg[1], "own": None}[_COMM_STREAM]
            self.reducer = GradReducer(eng.flat_grad, eng.live_names, eng.offsets, comm_stream=comm_stream)
            optimiser.grad_scale = self.reducer.grad_scale

    def __call__(self, batch):
        """batch: dict with the reference schema (image [B,3,H,W] + six [B,H,W] label maps), already on the GPU."""
        if self.use_plan:
            return self._planned(batch)
        if not self.use_graph:
            return self._eager(batch, None)
        if self._graph is not None and all(batch[k].shape == v.shape for k, v in self._static.items()):
            for k, v in self._static.items():
                if batch[k].data_ptr() != v.data_ptr():
                    v.copy_(batch[k], non_blocking=True)
            self._hyper.copy_(self.optimiser.next_hyper(), non_blocking=True)
            self._graph.replay()
            self.eng.weights_dirty = True                           # for eager forwards (evaluation) between replays
            return self.losses
        if self._graph is not None:                                 # new batch shape: start over
            self._graph, self._eager_steps = None, 0
        if self._eager_steps < 2:                                   # warm-up: arena / workspace allocations, packing tables
            self._eager_steps += 1
            return self._eager(batch, None)
        # capture: same schedule, Adam scalars from device memory
        self._static = {k: v.clone() for k, v in batch.items() if torch.is_tensor(v)}
        self._hyper = torch.zeros(7, device=self.eng.device)
        self._hyper.copy_(self.optimiser.next_hyper())
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._eager(self._static, self._hyper)
        self._graph = g
        g.replay()                                                  # capture does not execute: run the step it recorded
        self.eng.weights_dirty = True
        return self.losses

    # ---- recorded launch plan --------------------------------------------------------------------------------------------
    def _planned(self, batch):
        from .. import _lib
        lib = _lib.load()
        shape = tuple(batch["image"].shape)
        if shape != self._plan_shape:                     # new batch shape: the arena / outputs are re-allocated -> every plan is stale
            self._drop_plans()
            self._plan_shape, self._eager_steps = shape, 0
        if self._plans and self._plan_gen != ops.alloc_generation():
            # something a plan addresses by raw pointer was re-allocated since it was recorded (a validation forward on a larger
            # batch grew the arena or a workspace, the parameters were re-flattened, a pack table was rebuilt): replaying would
            # read and write freed memory.  Start over: two eager steps, then record again.
            self._drop_plans()
            self._eager_steps = 0
        # a plan bakes the pointers the kernels read: only inputs the engine consumes in place qualify (a non-contiguous or
        # non-fp32 tensor is converted into a temporary whose address dies with the step) -- those batches stay eager
        if not all(batch[k].is_contiguous() and batch[k].dtype == torch.float32 for k in ("image",) + TARGET_KEYS):
            return self._eager(batch, None)
        if self._eager_steps < 2:                         # warm-up: arena, workspaces, packing tables reach their final addresses
            self._eager_steps += 1
            return self._eager(batch, None)
        key = tuple(batch[k].data_ptr() for k in ("image",) + TARGET_KEYS)
        rec = self._plans.get(key)
        if rec is None:
            if len(self._plans) >= 8:                     # inputs that never repeat (no static / double-buffered batches): stay eager
                return self._eager(batch, None)
            gen0 = ops.alloc_generation()
            rec = self._record_plan(batch, lib, _lib)
            if ops.alloc_generation() != gen0:            # the recording step itself still allocated: its plan holds dead addresses
                _lib.load().fp_plan_destroy(rec[0])
                self._drop_plans()
                return self.losses
            self._plans[key] = rec
            self._plan_gen = gen0
            return self.losses
        plan, marks, adam_at, n = rec
        self._hyper.copy_(self.optimiser.next_hyper(), non_blocking=True)
        with ops.on_stream(torch.cuda.current_stream()):
            pos = 0
            if self.reducer is not None and not self.reducer.plan_recordable:      # framework collectives: replay in pieces around them
                for mark, stage in marks:
                    _lib.check(lib.fp_plan_replay(plan, pos, mark), "fp_plan_replay")
                    self.reducer.stage_ready(stage, self.eng.stage_streams())
                    pos = mark
                _lib.check(lib.fp_plan_replay(plan, pos, adam_at), "fp_plan_replay")
                self.reducer.finish()
                pos = adam_at
            _lib.check(lib.fp_plan_replay(plan, pos, n), "fp_plan_replay")
        self.eng.weights_dirty = True                     # for eager forwards (evaluation) between replays
        return self.losses

    def _record_plan(self, batch, lib, _lib):
        """one real step, recorded: every kernel launch and stream-ordering edge the library issues goes into the plan"""
        if self._hyper is None:
            self._hyper = torch.zeros(7, device=self.eng.device)
        self._hyper.copy_(self.optimiser.next_hyper())
        self.eng.weights_dirty = True                     # the weight repack must be part of every plan
        torch.cuda.synchronize()
        plan = lib.fp_plan_begin()
        if not plan:
            raise RuntimeError("fp_plan_begin: a plan is already recording on this thread")
        marks, adam_at = [], [0]
        try:
            self._eager(batch, self._hyper, on_mark=lambda stage: marks.append((lib.fp_plan_mark(plan), stage)),
                        before_adam=lambda: adam_at.__setitem__(0, lib.fp_plan_mark(plan)))
        finally:
            n = lib.fp_plan_end(plan)
        if n <= 0:
            raise RuntimeError("fp_plan_end: %s" % lib.fp_last_error_string().decode())
        return plan, marks, adam_at[0], n

    def _drop_plans(self):
        if self._plans:
            from .. import _lib
            torch.cuda.synchronize()
            for plan, *_ in self._plans.values():
                _lib.load().fp_plan_destroy(plan)
            self._plans = {}

    def __del__(self):
        try:
            self._drop_plans()
        except Exception:
            pass

    def _eager(self, batch, hyper_dev, on_mark=None, before_adam=None):
        eng = self.eng
        img = batch["image"]
        if self.outputs is None or self.outputs[0].shape[0] != img.shape[0] or self.outputs[0].shape[2:] != img.shape[2:]:
            B, _, H, W = img.shape
            self.outputs = [torch.empty((B, 4, H, W), device=eng.device) for _ in range(4)]
            self.dpreds = [torch.empty((B, 4, H, W), device=eng.device) for _ in range(4)]
        if not self.model.training:
            self.model.train()
        eng.forward(img, training=True, save_for_backward=True, outputs=self.outputs)
        ops.loss_fwd_bwd(self.outputs, batch, self.losses, self.dpreds, self.depth_range, self.prior)
        # zero_grad + backward: gradients are overwritten; buckets are all-reduced as soon as they are complete
        on_stage = None
        staged, armed = [], False                         # stages whose slice of the parameters was updated under the backward pass
        if self.reducer is not None and self.reducer.overlap:
            def on_stage(stage):                          # recording: remember where the stage's gradients are complete
                if on_mark is not None:
                    on_mark(stage)
                self.reducer.stage_ready(stage, eng.stage_streams())
        elif self.reducer is None and self._adam_ranges is not None and eng.stage_streams() is not None:
            armed = True
            opt, side = self.optimiser, eng.dwg[0]        # the mask decoder's weight-gradient stream: idle for most of the encoder's backward
            if hyper_dev is None:
                opt.begin_fused_step(eng)
            else:
                opt._buffers(eng)

            def on_stage(stage):
                lo, hi = self._adam_ranges.get(stage, (0, 0))
                if hi - lo < _ADAM_STAGE_MIN or stage in staged:
                    return
                # nothing later in the step reads these parameters: the convolutions use the packed copies of the step's start, and a
                # stage's BatchNorm backward (which reads gamma in place) has been launched on a stream waited for here
                for st in eng.stage_streams():
                    if st.cuda_stream != side.cuda_stream:
                        ops.event_wait(side, ops.event_record(st))
                with ops.on_stream(side):
                    opt.fused_range(eng, lo, hi, hyper_dev)
                staged.append(stage)
        eng.backward(self.dpreds, accumulate=False, on_stage=on_stage)
        if before_adam is not None:
            before_adam()
        if self.reducer is not None:
            self.reducer.finish()
        if armed:
            # the rest in flat order (adjacent stages merge into one launch); the step ends when the side stream's pieces have, too
            for lo, hi in _complement([self._adam_ranges[s] for s in staged], eng.flat_grad.numel()):
                self.optimiser.fused_range(eng, lo, hi, hyper_dev)
            if staged:
                ops.stream_wait_stream(ops.current_stream(), eng.dwg[0])
        elif hyper_dev is None:
            self.optimiser.fused_step(eng)
        else:
            self.optimiser.graph_step(eng, hyper_dev)
        return self.losses

    def losses_dict(self):
        host = self.losses.cpu()
        return {k: float(host[i]) for i, k in enumerate(LOSS_KEYS)}


def _complement(ranges, total):
    """[lo, hi) pieces of [0, total) not covered by `ranges` (disjoint), in ascending order"""
    out, p = [], 0
    for lo, hi in sorted(ranges):
        if lo > p:
            out.append((p, lo))
        p = max(p, hi)
    if p < total:
        out.append((p, total))
    return out


def synthetic_batch(B, H, W, device, seed=SEED):
    """Synthetic batch with the reference schema and the distributions of SURVEY.md section 8(d)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g)
    t = {"image": r(B, 3, H, W)}
    t["visible_ground"] = (r(B, H, W) < 0.4).float()
    t["depth"] = r(B, H, W) * 80.0 * (r(B, H, W) < 0.8).float()
    t["ground_depth"] = r(B, H, W) * 30.0 * (r(B, H, W) < 0.5).float()
    t["moving_object_mask"] = (r(B, H, W) < 0.05).float()
    t["depth_mask"] = (r(B, H, W) < 0.1).float()
    t["all_ground"] = ((t["ground_depth"] + t["visible_ground"]) > 0).float()
    return {k: v.to(device) for k, v in t.items()}


class SyntheticLoader:
    """`steps` batches per epoch with the reference schema (datasets/footprint_dataset.py:55-65): a small pool of synthetic
    batches resident on the device, cycled.  Stands in for the KITTI / Matterport DataLoaders, which are out of scope."""

    def __init__(self, batch_size, height, width, steps, seed=SEED, pool=2, device="cuda"):
        self.steps, self.first, self.stride = steps, 0, 1
        self.batches = [synthetic_batch(batch_size, height, width, device, seed=seed + i) for i in range(pool)]
        self.dataset = range(steps * batch_size)           # len(loader.dataset) is what the reference logs (train.py:76-77)

    def __len__(self):
        return len(range(self.first, self.steps, self.stride))

    def __iter__(self):
        for i in range(self.first, self.steps, self.stride):
            yield dict(self.batches[i % len(self.batches)])

    def shard(self, rank, world):
        """the per-rank view parallel.ShardedLoader asks for: batches rank, rank + world, ... of the same pool, equal counts on every rank"""
        import copy
        v = copy.copy(self)
        v.steps = (self.steps // world) * world
        v.first, v.stride = rank, world
        return v


class _Opts:
    """keyword-style construction of the options namespace (the reference passes an argparse namespace, options.py)"""

    def __init__(self, **kw):
        self.epochs, self.lr, self.depth_range, self.footprint_prior = 10, 1e-4, [0.1, 100], 0.25
        self.log_freq, self.val_batches, self.log_path, self.model_name, self.load_path = 250, 10, None, "model", None
        self.__dict__.update(kw)


class TrainManager:
    """Reference loop structure (training/train.py:42-215) over any iterable of batches with the reference schema.

    `TrainManager(options, train_loader=..., val_loader=...)` with the namespace of `footprints_amd.options.Options` (same
    flags as the reference), or `TrainManager(loader, epochs=..., ...)` keyword style.  Per step (train.py:147-159): the fused
    `TrainStep` (forward + loss + zero_grad + backward + Adam in one schedule) and the 21 losses handed to the `Evaluator` as
    one device vector (no synchronisation); every 100 steps the averaged-loss console line (train.py:161-166); every
    `log_freq` steps a validation pass over `val_batches` batches in eval mode (train.py:174-185, 193-215: forward + loss
    through the drop-in `model(x)` / `Evaluator.compute_losses` surface under no_grad); per epoch the checkpoint and
    `scheduler.step()` (train.py:189-191).  Tensorboard writing is out of scope: `self.history` keeps
    what `log(...)` would have received.

    Data-parallel training (north_star config #4; the reference is single-process): started through `python -m torch.distributed.run
    --nproc-per-node N -m footprints_amd.main ...`, every rank builds the same manager on its own GPU; `DistContext` (parallel.py)
    shards the loader, `broadcast_state` makes the replicas identical, `TrainStep(distributed=True)` all-reduces the gradient buckets
    over RCCL under the backward pass, rank 0 alone prints and saves, logged losses are averaged over the ranks."""

    def __init__(self, options=None, train_loader=None, val_loader=None, log=print, dist_context=None, model_manager=None,
                 train_step=None, **kw):
        if options is not None and not hasattr(options, "epochs"):       # keyword style: first positional is the loader
            train_loader, options = options, None
        self.opt = options if options is not None else _Opts(**kw)
        if options is not None and kw:
            self.opt.__dict__.update(kw)
        if "learning_rate" in kw:
            self.opt.lr = kw["learning_rate"]
        if "loader" in kw:
            train_loader = kw["loader"]
        torch.manual_seed(SEED)                                          # train.py:33-35
        # data-parallel context (launched through torch.distributed.run: WORLD_SIZE > 1 in the environment): one process per GPU, the
        # device is selected BEFORE the model is built; rank r trains on every world-th batch of the loader, only rank 0 talks and saves
        self.dist = dist_context if dist_context is not None else DistContext.from_env()
        self.train_loader, self.val_loader = self.dist.shard(train_loader), val_loader
        self.log = log if self.dist.is_main else (lambda *a, **k: None)
        self.loader = self.train_loader
        save_folder = kw.get("save_folder")
        if save_folder is None and getattr(self.opt, "log_path", None):
            save_folder = os.path.join(self.opt.log_path, self.opt.model_name, "models")     # train.py:57-59
        # (model_manager / train_step can be injected: the CPU tests drive this loop with stand-ins, there is no CPU compute path)
        self.model_manager = model_manager if model_manager is not None else ModelManager(
            save_folder=save_folder, use_cuda=True, learning_rate=self.opt.lr, lr_step_size=kw.get("lr_step_size", 10))
        if getattr(self.opt, "load_path", None) is not None:
            self.model_manager.load_model(weights_path=self.opt.load_path, load_optimiser=True)   # train.py:63-64
        self.model = self.model_manager.model
        self.optimiser, self.scheduler = self.model_manager.optimiser, self.model_manager.scheduler
        self.val_iter = iter(self.val_loader) if self.val_loader is not None else None
        depth_range = tuple(self.opt.depth_range)
        self.evaluator = Evaluator(depth_range, self.opt.footprint_prior, compute_viz=False)
        if self.dist.active and train_step is None:
            broadcast_state(self.model, src=0, group=self.dist.group)     # every replica starts from rank 0's weights, buffers and Adam-free state
        self.train_step = train_step if train_step is not None else TrainStep(
            self.model, self.optimiser, depth_range, self.opt.footprint_prior, distributed=self.dist.active)
        self.epochs = self.opt.epochs
        self.step = 0
        self.lr = self.opt.lr
        self.num_total_steps = (len(train_loader) if hasattr(train_loader, "__len__") else 0) * self.epochs
        self.train_network_time = self.val_time = 0.0
        self.history = {"train": [], "val": []}                          # (step, averaged losses) per log event

    def train(self):
        self.start_time = time.time()
        for self.epoch in range(self.epochs):
            self.run_epoch()

    def run_epoch(self):
        for batch_idx, inputs in enumerate(self.train_loader):
            t0 = time.time()
            if torch.cuda.is_available():
                inputs = {k: v.cuda(non_blocking=True) for k, v in inputs.items()}     # process_batch: train.py:220-222
            losses = self.train_step(inputs)                                       # train.py:150-156 in one schedule
            self.evaluator.add_loss_vector(losses, mode="train")
            self.lr = self.scheduler.get_last_lr()[0]
            self.train_network_time += time.time() - t0
            if self.step % 100 == 0:                                               # train.py:161-185
                # data-parallel: the logged numbers are the mean over the ranks' shards (one 21-float all-reduce per log event,
                # entered by every rank -- the step counter is the same everywhere)
                avg = self.dist.mean_losses(self.evaluator.get_averaged_losses(mode="train", reset=False))
                self.log("Epoch {} -- Batch {} -- Loss {}".format(self.epoch, batch_idx, avg["loss"]))
                if self.step % self.opt.log_freq == 0:
                    avg = self.dist.mean_losses(self.evaluator.get_averaged_losses(mode="train", reset=True))
                    self.history["train"].append((self.step, avg))
                    if self.val_loader is not None:
                        self.model.eval()
                        self.val()
                        self.model.train()
            self.step += 1
        # checkpoint: rank 0 only (weights are bit-identical on every rank after each step; BatchNorm running statistics are per
        # replica and rank 0's are the ones kept -- SURVEY.md section 8e); the others wait so that nobody races ahead into a load
        if self.model_manager.save_folder is not None and self.dist.is_main:
            self.model_manager.save_model(folder_name="weights_{}".format(self.epoch))   # train.py:190
        self.dist.barrier()
        self.scheduler.step()                                                             # train.py:191

    def val(self):
        """train.py:193-215 (with `next(it)`: the reference's `self.val_iter.next()` no longer exists in PyTorch)"""
        t0 = time.time()
        with torch.no_grad():
            for _ in range(self.opt.val_batches):
                try:
                    inputs = next(self.val_iter)
                except StopIteration:
                    self.val_iter = iter(self.val_loader)
                    inputs = next(self.val_iter)
                self.process_batch(inputs, mode="val", return_batch_loss=False)
        avg = self.dist.mean_losses(self.evaluator.get_averaged_losses(mode="val", reset=True))
        self.history["val"].append((self.step, avg))
        self.val_time += time.time() - t0
        return avg

    def process_batch(self, inputs, mode="train", return_batch_loss=False):
        """train.py:217-227: the drop-in surface -- model(x) then Evaluator.compute_losses"""
        if torch.cuda.is_available():
            inputs = {k: v.cuda(non_blocking=True) for k, v in inputs.items()}
        outputs = self.model(inputs["image"])
        losses = self.evaluator.compute_losses(inputs, outputs, mode=mode, return_batch_loss=return_batch_loss)
        return outputs, losses
