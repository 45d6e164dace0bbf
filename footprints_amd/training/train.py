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
from ..parallel import GradReducer
from .evaluation import Evaluator
from .losses import LOSS_KEYS, SCALES, TARGET_KEYS

SEED = 10   # train.py:33


_GRAPH = bool(int(os.environ.get("FP_GRAPH", "0")))     # hipGraph replay of the whole step (single-GPU TrainStep); opt-in, see below


class TrainStep:
    """One full training step (forward + fused loss + backward + Adam) on static buffers, no autograd graph.

    graph=True (or FP_GRAPH=1; single-GPU only): after two eager steps the whole step -- ~770 launches on five streams -- is
    captured once into a hipGraph and replayed; the batch is copied into static input buffers and the seven Adam scalars of the
    step are uploaded before each replay, so results are bit-identical to eager (tested).  It is OFF by default: measured on
    ROCm 7.2 / MI355X the replay costs the host as much as the eager launches (16.4 vs 14.9 ms per step), does not shorten the
    dependent-launch gaps (one stream: 497 img/s both ways) and runs the forked streams with less overlap (509 vs 578 img/s)."""

    def __init__(self, model, optimiser, depth_range=(0.1, 100.0), footprint_prior=0.25, distributed=False, graph=None):
        self.model, self.optimiser = model, optimiser
        self.use_graph = (_GRAPH and not distributed) if graph is None else bool(graph)
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
        if distributed:
            self.reducer = GradReducer(self.eng.flat_grad, self.eng.live_names, self.eng.offsets)
            optimiser.grad_scale = self.reducer.grad_scale

    def __call__(self, batch):
        """batch: dict with the reference schema (image [B,3,H,W] + six [B,H,W] label maps), already on the GPU."""
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

    def _eager(self, batch, hyper_dev):
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
        eng.backward(self.dpreds, accumulate=False, on_stage=self.reducer.stage_ready if (self.reducer is not None and self.reducer.overlap) else None)
        if self.reducer is not None:
            self.reducer.finish()
        if hyper_dev is None:
            self.optimiser.fused_step(eng)
        else:
            self.optimiser.graph_step(eng, hyper_dev)
        return self.losses

    def losses_dict(self):
        host = self.losses.cpu()
        return {k: float(host[i]) for i, k in enumerate(LOSS_KEYS)}


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


class TrainManager:
    """Reference loop structure (train.py:42-227) over any iterable of batches."""

    def __init__(self, loader, epochs=10, learning_rate=1e-4, lr_step_size=10, depth_range=(0.1, 100.0), footprint_prior=0.25,
                 save_folder=None, log=print):
        torch.manual_seed(SEED)
        self.loader, self.epochs, self.log = loader, epochs, log
        self.model_manager = ModelManager(save_folder=save_folder, use_cuda=True, learning_rate=learning_rate,
                                          lr_step_size=lr_step_size)
        self.model = self.model_manager.model
        self.optimiser, self.scheduler = self.model_manager.optimiser, self.model_manager.scheduler
        self.evaluator = Evaluator(depth_range, footprint_prior, compute_viz=False)
        self.train_step = TrainStep(self.model, self.optimiser, depth_range, footprint_prior)
        self.step = 0
        self.train_network_time = 0.0

    def train(self):
        for self.epoch in range(self.epochs):
            self.run_epoch()

    def run_epoch(self):
        for batch_idx, inputs in enumerate(self.loader):
            t0 = time.time()
            inputs = {k: v.cuda(non_blocking=True) for k, v in inputs.items()}     # train.py:220-222
            losses = self.train_step(inputs)
            if self.step % 100 == 0:
                torch.cuda.synchronize()
                self.log("Epoch {} -- Batch {} -- Loss {}".format(self.epoch, batch_idx, float(losses[20])))
            self.train_network_time += time.time() - t0
            self.step += 1
        if self.model_manager.save_folder is not None:
            self.model_manager.save_model(folder_name="weights_{}".format(self.epoch))   # train.py:190
        self.scheduler.step()                                                             # train.py:191
