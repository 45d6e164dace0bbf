"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI, overlapped with backward.

The reference is single-GPU; this is the new N-GPU path of BASELINE.json's north_star.
Semantics: every rank runs the same step on its own per-GPU batch with per-replica
BatchNorm statistics; the live gradients (31 012 944 fp32 = 124 MB, one flat buffer) are SUMMED across ranks
and the 1/world factor is folded into the fused Adam update (grad_scale), so all ranks hold bit-identical
weights after every step.  The flat gradient buffer is cut into contiguous buckets in the order backward
produces them (mask decoder, depth decoder, encoder layer4 .. layer0); each bucket's all-reduce is issued on a
side stream as soon as the engine reports it ready, so communication hides under the remaining backward.
`torch.distributed` (backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests) is plumbing only.
"""
import os

import torch
import torch.distributed as dist

# The bucket all-reduces overlap the remaining backward by default (north_star).  Round 1 kept them behind the backward pass
# because RCCL's gfx950 code object contains packed-fp32 VALU instructions and a packed-fp32 kernel of this library
# (head_wgrad, SLP-vectorised build) had been seen to return different sums next to the bf16-MFMA convolution.  Round 2 measured
# (scripts/ubench/pk_hazard.hip, profiles/round2_notes.md): four standalone victims -- v_pk_fma_f32 accumulation, the
# reduction-kernel pattern out = a + b with v_pk_add_f32, the compiler-packed op_sel forms of head_wgrad's own loop with its
# shuffle tree -- are bit-stable next to a resident bf16-MFMA spinner (0 of 6 runs differ, each), so packed fp32 VALU beside
# MFMA is not a hazard by itself; the head_wgrad effect reproduces only with that one kernel built with the SLP vectoriser (the
# library stays built without it) and has nothing to do with collectives.  FP_DP_OVERLAP=0 reduces after the backward pass.
_OVERLAP = bool(int(os.environ.get("FP_DP_OVERLAP", "1")))
# FP_DP_FORCE=1: issue the bucket collectives even in a world of one rank (a 1-GPU box can then execute the RCCL code path:
# communicator creation, bucket all-reduces on the comm stream, event gating -- tests/test_gpu_dp.py, `bench.py --force-dist`)
_FORCE = bool(int(os.environ.get("FP_DP_FORCE", "0")))


def bucket_ranges(names, offsets, total, max_elems=8 << 20):
    """Contiguous [lo, hi) ranges of the flat buffer, grouped by top-level stage, listed in backward order.

    Stages in forward/flat order: encoder.layer0..4, mask_decoder, depth_decoder.  Large stages are further cut
    into <= max_elems pieces so the first all-reduce can start early and ring steps stay pipelined.
    """
    def stage(n):
        parts = n.split(".")
        return parts[0] + "." + parts[1] if parts[0] == "encoder" else parts[0]

    stages = []
    for n, o in zip(names, offsets):
        s = stage(n)
        if not stages or stages[-1][0] != s:
            stages.append([s, o, o])
    for i, st in enumerate(stages):
        st[2] = stages[i + 1][1] if i + 1 < len(stages) else total
    by_name = {s: (lo, hi) for s, lo, hi in stages}
    order = ["mask_decoder", "depth_decoder", "encoder.layer4", "encoder.layer3", "encoder.layer2", "encoder.layer1", "encoder.layer0"]
    out = []
    for s in order:
        lo, hi = by_name[s]
        # backward produces a stage's gradients from its END towards its start -> emit pieces high to low
        pieces = []
        p = hi
        while p > lo:
            q = max(lo, p - max_elems)
            pieces.append((s, q, p))
            p = q
        out.extend(pieces)
    return out


class GradReducer:
    """Bucketed, stream-overlapped all-reduce of a flat gradient buffer."""

    def __init__(self, flat_grad, names, offsets, group=None, max_elems=8 << 20, overlap=None, force=None):
        self.flat = flat_grad
        self.force = _FORCE if force is None else bool(force)
        # overlap: issue each bucket as soon as its stage is complete (default on CPU/gloo; on GPUs see _OVERLAP above)
        self.overlap = (_OVERLAP or not flat_grad.is_cuda) if overlap is None else bool(overlap)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.buckets = bucket_ranges(names, offsets, flat_grad.numel(), max_elems)
        self.cuda = flat_grad.is_cuda
        self.stream = torch.cuda.Stream() if self.cuda else None
        self._pending = []
        self._done_stage = set()

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def stage_ready(self, stage, streams=None):
        """Called by the backward schedule when every gradient of `stage` has been launched.  streams: every stream that may still be
        writing them (Engine.stage_streams(): the weight gradients run on side streams that the main stream does not wait for); the
        collective is ordered after all of them.  None: the current stream only."""
        if (self.world == 1 and not self.force) or stage in self._done_stage:
            return
        self._done_stage.add(stage)
        if self.cuda:
            for st in (streams or [torch.cuda.current_stream()]):
                ev = torch.cuda.Event()
                ev.record(st)
                self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                for s, lo, hi in self.buckets:
                    if s == stage:
                        self._pending.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            for s, lo, hi in self.buckets:
                if s == stage:
                    dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group)

    def finish(self):
        """Make the compute stream wait for every outstanding bucket (call before the optimiser step)."""
        if self.world > 1 or self.force:
            for s in ("mask_decoder", "depth_decoder", "encoder.layer4", "encoder.layer3", "encoder.layer2", "encoder.layer1",
                      "encoder.layer0"):
                self.stage_ready(s)          # anything the schedule did not report explicitly (the backward pass has joined its streams by now)
            for w in self._pending:
                w.wait()
            if self.cuda:
                torch.cuda.current_stream().wait_stream(self.stream)
        self._pending = []
        self._done_stage = set()


def broadcast_state(model, src=0, group=None):
    """Initial weight/buffer sync so every replica starts identical (rank `src` wins).  With a HIP engine attached the live
    parameters are ONE flat buffer: broadcast that (one collective instead of 196), then tell the engine its packed / BN-folded
    weight copies are stale -- `.data` writes do not move `Parameter._version` (Engine.invalidate)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    eng = getattr(model, "_engine", None)
    done = set()
    if eng is not None and eng.params_alias_flat():
        dist.broadcast(eng.flat_param, src=src, group=group)
        done = {id(p) for p in eng.live_params}
    for t in list(model.parameters()) + list(model.buffers()):
        if id(t) not in done:
            dist.broadcast(t.data, src=src, group=group)
    if eng is not None:
        eng.invalidate()
