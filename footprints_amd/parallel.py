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


# Transport of the bucket collectives on GPUs: "rccl" = this library's own fp_comm_* entry points (csrc/comm.cpp: ncclAllReduce on
# a stream the engine names, recordable into a launch plan, no framework watchdog / side streams), "torch" = torch.distributed
# collectives on the group (gloo in the CPU tests and when two test ranks share one GPU; "nccl" as the fallback when fp_comm_init
# fails on some rank).  FP_DP_TRANSPORT=torch|rccl overrides the choice.
_TRANSPORT = os.environ.get("FP_DP_TRANSPORT", "")
_ALLOW_SHARED = bool(int(os.environ.get("FP_DP_ALLOW_SHARED_GPU", "0")))     # several ranks on one physical GPU (gloo transport): tests only
# which stream carries the all-reduces: "own" = a dedicated stream; "dwg0" / "dwg1" / "aux" / "wg" = that engine stream (ROCm maps a
# process's streams onto a handful of hardware queues -- every extra stream that is busy during the backward pass can end up sharing
# a queue with one of the five the schedule already uses, and streams that share a queue serialise)
_COMM_STREAM = os.environ.get("FP_DP_COMM_STREAM", "dwg0")


class Communicator:
    """RCCL communicator behind the C ABI (fp_comm_*): one per process, on the current device.  The 128-byte unique id travels from
    rank 0 over the torch.distributed group (any backend; object broadcast) -- host plumbing only."""

    def __init__(self, group=None):
        from . import _lib
        import ctypes as C
        self._lib, self._C = _lib, C
        lib = _lib.load()
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        nbytes = lib.fp_comm_unique_id_bytes()
        ident, err = [None], None
        if self.rank == 0:
            try:
                raw = C.create_string_buffer(nbytes)
                _lib.check(lib.fp_comm_unique_id(raw, nbytes), "fp_comm_unique_id")
                ident = [raw.raw]
            except Exception as e:       # the other ranks are already waiting in the broadcast: let them see the failure too
                err = e
        if self.world > 1:
            dist.broadcast_object_list(ident, src=0, group=group)
        if ident[0] is None:
            raise err if err is not None else RuntimeError("rank 0 could not create an RCCL unique id")
        handle = C.c_void_p()
        _lib.check(lib.fp_comm_init(ident[0], self.rank, self.world, C.byref(handle)), "fp_comm_init")
        self.handle = handle

    def count(self):
        """ranks of the communicator as RCCL itself counts them (ncclCommCount)"""
        return int(self._lib.load().fp_comm_count(self.handle)) if self.handle else 0

    def allreduce(self, t, stream):
        """in-place sum over the ranks, asynchronous on `stream` (a torch stream)"""
        self._lib.check(self._lib.load().fp_comm_allreduce_async(self.handle, t.data_ptr(), t.numel(), stream.cuda_stream),
                        "fp_comm_allreduce_async")

    def broadcast(self, t, root, stream):
        self._lib.check(self._lib.load().fp_comm_broadcast(self.handle, t.data_ptr(), t.numel(), root, stream.cuda_stream),
                        "fp_comm_broadcast")

    def destroy(self):
        if self.handle:
            # recorded launch plans hold this communicator by raw pointer (csrc/plan.cpp NODE_ALLREDUCE): make every TrainStep drop its plans
            # before one could be replayed on the freed handle
            from . import ops
            ops.bump_alloc_generation()
            self._lib.load().fp_comm_destroy(self.handle)
            self.handle = None


_COMMS = {}          # group (None = the default group) -> Communicator or None; the key keeps the group object alive (id() values are recycled)
_HOST_GROUPS = {}


def _host_group(group):
    """a group whose collectives run on the host (gloo): `group` itself when it is gloo-backed, else a gloo twin created once by all ranks
    together -- the few bytes of host plumbing (flags, device census, logged losses) must not instantiate the framework's own NCCL
    communicator with its watchdog thread and streams (they would take hardware queues from the engine, profiles/round3_notes.md)"""
    if "gloo" in str(dist.get_backend(group)):
        return group
    if group not in _HOST_GROUPS:
        # dist.new_group must be entered by ALL ranks of the default group: created lazily from inside a sub-group's collective, the
        # non-members would never call it and the members would wait forever.  The default group is always complete here;
        # a non-gloo SUB-group needs its twin made up front by everybody: prepare_host_group(group) on every rank right after the group
        # is created (DistContext.from_env does it for the default group; sub-groups are the caller's)
        if group is not None:
            raise RuntimeError("footprints_amd.parallel: the gloo twin of a non-gloo sub-group has to be created by all ranks of the default "
                               "group together: call parallel.prepare_host_group(group) on every rank right after creating the group")
        _HOST_GROUPS[group] = dist.new_group(ranks=None, backend="gloo")
    return _HOST_GROUPS[group]


def prepare_host_group(group=None):
    """create the gloo twin of `group` NOW; every rank of the DEFAULT group has to call this (members and non-members alike)"""
    if "gloo" in str(dist.get_backend(group)) or group in _HOST_GROUPS:
        return
    ranks = dist.get_process_group_ranks(group) if group is not None else None
    _HOST_GROUPS[group] = dist.new_group(ranks=ranks, backend="gloo")


def get_communicator(group=None, create=True):
    """the process's fp_comm communicator for `group`, created on first use by ALL ranks together; None when it cannot be used: every
    rank reports whether its fp_comm_init succeeded and the transport is only chosen if all did (else everyone falls back to
    torch.distributed collectives on the group)."""
    key = group
    if key in _COMMS or not create:
        return _COMMS.get(key)
    multi = dist.is_initialized() and dist.get_world_size(group) > 1

    def agree(ok):                # MIN over the ranks, on the host
        if not multi:
            return ok
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=_host_group(group))
        return int(flag.item())

    def complain(e):
        print("footprints_amd.parallel: fp_comm unavailable on rank %s (%s); falling back to torch.distributed collectives"
              % (dist.get_rank(group) if dist.is_initialized() else 0, e), flush=True)

    # pre-flight: ncclCommInitRank blocks until EVERY rank has called it, so a rank that cannot even load RCCL must be found out before
    # anybody enters it
    comm, ok = None, 1
    try:
        from . import _lib
        if _lib.load().fp_comm_version() < 0:
            raise RuntimeError("librccl could not be loaded: " + _lib.load().fp_last_error_string().decode())
    except Exception as e:
        ok = 0
        complain(e)
    if not agree(ok):
        _COMMS[key] = None
        return None
    try:
        comm = Communicator(group)
    except Exception as e:        # bootstrap failure, ...: agree on the fallback below, loudly
        ok = 0
        complain(e)
    ok = agree(ok)
    if not ok and comm is not None:
        comm.destroy()
        comm = None
    _COMMS[key] = comm
    return comm


def destroy_communicators():
    for c in _COMMS.values():
        if c is not None:
            c.destroy()
    _COMMS.clear()
    _SHARED_DEVICE.clear()
    for twin in _HOST_GROUPS.values():          # the gloo twins this module created are its own to destroy; a re-initialised
        try:                                    # process group must not find the twin of the destroyed one either
            if dist.is_initialized():
                dist.destroy_process_group(twin)
        except Exception:                       # already gone with the default group
            pass
    _HOST_GROUPS.clear()


_SHARED_DEVICE = {}


def _physical_device_id(index=None):
    """identity of the PHYSICAL GPU behind the current logical device: its uuid, else its PCI address.  The logical index is not enough
    (ADVICE r4): launchers that bind one GPU per rank through HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES (SLURM --gpus-per-task, some
    k8s device plugins) give EVERY rank logical device 0 -- a census of (host, 0) would call that "all ranks share one GPU" and downgrade a
    healthy 8-GPU run to gloo."""
    index = torch.cuda.current_device() if index is None else index
    p = torch.cuda.get_device_properties(index)
    uuid = getattr(p, "uuid", None)
    pci = tuple(getattr(p, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    have_pci = any(v is not None for v in pci)
    if uuid is not None and str(uuid).strip("0-") != "":
        # partitions of one GPU (CPX / NPS modes) may report the same uuid but sit on PCI functions of their own: both go into the identity
        # so that they do not read as "two ranks on one device" 
        return "uuid:%s%s" % (uuid, "/pci:%s" % (pci,) if have_pci else "")
    if have_pci:
        return "pci:%s" % (pci,)
    vis = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES", "")))
    return "logical:%s/%d" % (vis, index)


def ranks_share_a_device(group=None):
    """True when two ranks of the group sit on the same GPU of the same host (RCCL refuses that: the 1-GPU test box).  A collective over
    the host group on first use -- every rank calls it at the same points (GradReducer / broadcast_state construction)."""
    if group not in _SHARED_DEVICE:
        import socket
        mine = (socket.gethostname(), _physical_device_id())
        everyone = [None] * dist.get_world_size(group)
        dist.all_gather_object(everyone, mine, group=_host_group(group))
        _SHARED_DEVICE[group] = len(set(everyone)) < len(everyone)
    return _SHARED_DEVICE[group]


def _pick_transport(flat_grad, group, force):
    """"rccl" (the library's fp_comm_* entry points) whenever every rank has a GPU of its own -- whatever backend the torch.distributed
    group uses for its host plumbing; "torch" collectives on the group for CPU tensors and for ranks that share a device, and that choice
    is announced: gloo stages every bucket through the host."""
    if not flat_grad.is_cuda:
        return "torch"
    if _TRANSPORT in ("torch", "rccl"):
        return _TRANSPORT
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return "rccl" if (force or dist.is_initialized()) else "torch"
    if ranks_share_a_device(group):
        # RCCL needs a device per rank.  Ranks sharing a GPU is a TEST configuration (the one-GPU test box): it has to be asked for
        # (FP_DP_ALLOW_SHARED_GPU=1 or FP_DP_TRANSPORT=torch), a production launch that ends up here by a binding mistake must not
        # silently train over gloo staged through the host at a fraction of the speed
        if not _ALLOW_SHARED:
            raise RuntimeError("footprints_amd.parallel: several ranks of the group sit on the same physical GPU; RCCL needs one device per "
                               "rank.  Fix the launcher's GPU binding, or set FP_DP_ALLOW_SHARED_GPU=1 (gradients then travel over "
                               "torch.distributed %s, staged through the host: a test configuration, not a data-parallel run)" % dist.get_backend(group))
        if dist.get_rank(group) == 0:
            print("footprints_amd.parallel: several ranks share one GPU (FP_DP_ALLOW_SHARED_GPU=1) -- gradients travel over "
                  "torch.distributed (%s), staged through the host.  This is a test configuration, not a data-parallel run."
                  % dist.get_backend(group), flush=True)
        return "torch"
    return "rccl"


class GradReducer:
    """Bucketed, stream-overlapped all-reduce of a flat gradient buffer."""

    def __init__(self, flat_grad, names, offsets, group=None, max_elems=8 << 20, overlap=None, force=None, comm_stream=None):
        self.flat = flat_grad
        self.force = _FORCE if force is None else bool(force)
        # overlap: issue each bucket as soon as its stage is complete (default on CPU/gloo; on GPUs see _OVERLAP above)
        self.overlap = (_OVERLAP or not flat_grad.is_cuda) if overlap is None else bool(overlap)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.buckets = bucket_ranges(names, offsets, flat_grad.numel(), max_elems)
        self.cuda = flat_grad.is_cuda
        self.active = self.world > 1 or self.force
        self.transport = _pick_transport(flat_grad, group, self.force) if self.active else "torch"
        self.comm = None
        if self.transport == "rccl":
            self.comm = get_communicator(group)
            if self.comm is None:
                self.transport = "torch"
        self.stream = (comm_stream if comm_stream is not None else torch.cuda.Stream()) if self.cuda else None
        self._pending = []
        self._done_stage = set()

    @property
    def grad_scale(self):
        return 1.0 / self.world

    @property
    def plan_recordable(self):
        """the collectives go through the library (fp_comm_allreduce_async) and become nodes of a recording launch plan"""
        return self.transport == "rccl"

    def stage_ready(self, stage, streams=None):
        """Called by the backward schedule when every gradient of `stage` has been launched.  streams: every stream that may still be
        writing them (Engine.stage_streams(): the weight gradients run on side streams that the main stream does not wait for); the
        collective is ordered after all of them.  None: the current stream only."""
        if not self.active or stage in self._done_stage:
            return
        self._done_stage.add(stage)
        mine = [(lo, hi) for s, lo, hi in self.buckets if s == stage]
        if not self.cuda:
            for lo, hi in mine:
                dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
            return
        from . import ops
        if self.transport == "rccl":
            for st in (streams or [ops.current_stream()]):
                if st.cuda_stream != self.stream.cuda_stream:
                    ops.event_wait(self.stream, ops.event_record(st))       # library events: a recording plan sees the edge
            for lo, hi in mine:
                self.comm.allreduce(self.flat[lo:hi], self.stream)
            return
        for st in (streams or [torch.cuda.current_stream()]):
            ev = torch.cuda.Event()
            ev.record(st)
            self.stream.wait_event(ev)
        with torch.cuda.stream(self.stream):
            for lo, hi in mine:
                self._pending.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Make the compute stream wait for every outstanding bucket (call before the optimiser step)."""
        if self.active:
            for s in ("mask_decoder", "depth_decoder", "encoder.layer4", "encoder.layer3", "encoder.layer2", "encoder.layer1",
                      "encoder.layer0"):
                self.stage_ready(s)          # anything the schedule did not report explicitly (the backward pass has joined its streams by now)
            for w in self._pending:
                w.wait()
            if self.cuda:
                if self.transport == "rccl":
                    from . import ops
                    if ops.current_stream().cuda_stream != self.stream.cuda_stream:
                        ops.stream_wait_stream(ops.current_stream(), self.stream)
                else:
                    torch.cuda.current_stream().wait_stream(self.stream)
        self._pending = []
        self._done_stage = set()


def broadcast_state(model, src=0, group=None):
    """Initial weight/buffer sync so every replica starts identical (rank `src` wins).  With a HIP engine attached the live
    parameters are ONE flat buffer: broadcast that (one collective instead of 196), then tell the engine its packed / BN-folded
    weight copies are stale -- `.data` writes do not move `Parameter._version` (Engine.invalidate)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    first = next(model.parameters())
    # the engine first: its streams must own their hardware queues before a communicator creates streams of its own (Engine.__init__)
    eng = model.engine() if (first.is_cuda and hasattr(model, "engine")) else getattr(model, "_engine", None)
    done = set()
    comm = get_communicator(group) if (first.is_cuda and _pick_transport(first.data, group, False) == "rccl") else None

    root = dist.get_group_rank(group, src) if group is not None else src      # `src` is a global rank; the communicator counts inside the group

    def bcast(t):
        if comm is None:
            dist.broadcast(t, src=src, group=group)
        elif t.dtype == torch.float32 and t.is_contiguous() and t.numel():
            comm.broadcast(t, root, torch.cuda.current_stream())
        elif "gloo" in str(dist.get_backend(group)):
            c = t.cpu()            # the few integer buffers (num_batches_tracked): through the host, so that the group's own CUDA
            dist.broadcast(c, src=src, group=group)      # backend (its communicator, watchdog and streams) is never instantiated
            t.copy_(c)
        else:
            dist.broadcast(t, src=src, group=group)
    if eng is not None and eng.params_alias_flat():
        bcast(eng.flat_param)
        done = {id(p) for p in eng.live_params}
    for t in list(model.parameters()) + list(model.buffers()):
        if id(t) not in done:
            bcast(t.data)
    if first.is_cuda:
        torch.cuda.current_stream().synchronize()
    if eng is not None:
        eng.invalidate()


class DistContext:
    """Host-side context of a data-parallel training run: rank / world from the launcher's environment
    (`python -m torch.distributed.run --nproc-per-node N -m footprints_amd.main ...`), one process per GPU.  torch.distributed is host
    plumbing only -- a gloo group for the rendezvous, the RCCL unique id, barriers and the 21 logged losses; gradients and the initial
    weights travel through the library's own communicator (fp_comm_*).  The reference trainer is single-process
    (footprints/training/train.py:42-215); everything here is what turns its loop into N replicas: per-rank shard of the loader,
    rank-0-only console line and checkpoint (eight ranks would race on `weights_{epoch}`, train.py:190), one all-reduce per log event."""

    def __init__(self, rank=0, world=1, local_rank=0, group=None):
        self.rank, self.world, self.local_rank, self.group = rank, world, local_rank, group

    @classmethod
    def from_env(cls, use_cuda=True):
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world <= 1 and not dist.is_initialized():
            return cls()
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if use_cuda and torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())      # before anything touches the GPU: one device per rank
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", init_method="env://", rank=int(os.environ.get("RANK", "0")), world_size=world)
        prepare_host_group(None)                # every rank is here: a non-gloo default group (launcher-made) gets its gloo twin now
        return cls(dist.get_rank(), dist.get_world_size(), local, None)

    @property
    def is_main(self):
        return self.rank == 0

    @property
    def active(self):
        return self.world > 1

    def shard(self, loader):
        return ShardedLoader(loader, self.rank, self.world, group=self.group) if self.active else loader

    def mean_losses(self, losses):
        """dict of floats -> the same keys averaged over the ranks (one small all-reduce on the host group; every rank calls it)"""
        if not self.active:
            return losses
        keys = list(losses)                 # the same insertion order on every rank (LOSS_KEYS)
        v = torch.tensor([float(losses[k]) for k in keys], dtype=torch.float64)
        dist.all_reduce(v, op=dist.ReduceOp.SUM, group=_host_group(self.group))
        v /= self.world
        return {k: float(x) for k, x in zip(keys, v)}

    def barrier(self):
        if self.active:
            dist.barrier(group=_host_group(self.group))


class ShardedLoader:
    """rank r's share of a loader's batches in a data-parallel run: every rank the same number of them (the tail that does not fill a round
    is dropped: a rank with one batch more would wait in an all-reduce nobody else enters).

    Sharding happens at the INDEX level wherever the wrapped loader allows it (round 5) -- a rank then pays the input pipeline
    (decode, augmentation, collation, H2D) for its own batches only, and disjointness does not depend on every rank drawing a bit-identical
    batch order:
      1. a loader with `shard(rank, world)` (SyntheticLoader, DeviceLoader / SyntheticSampleSource here) returns its own per-rank view:
         batches r, r + world, ... by index;
      2. a `torch.utils.data.DataLoader` is rebuilt over a rank-strided `EpochShardSampler` (epoch-seeded permutation shared by all ranks,
         `set_epoch` called from `__iter__`: the DistributedSampler contract, drop_last so that the ranks' counts agree);
      3. anything else is iterated completely by every rank, each keeping every world-th batch (the rounds 3-4 behaviour).  That is only
         correct when all ranks see the same order, so the first batch of every epoch is fingerprinted and the fingerprints are compared
         over the host group: ranks that disagree raise instead of silently training on overlapping shards."""

    def __init__(self, loader, rank, world, seed=0, group=None):
        self.loader, self.rank, self.world, self.group = loader, rank, world, group
        self.dataset = getattr(loader, "dataset", None)
        self.epoch = 0
        self.mode, self.inner = "stride", None
        if hasattr(loader, "shard") and callable(loader.shard):
            try:
                self.mode, self.inner = "index", loader.shard(rank, world)
            except (TypeError, NotImplementedError):
                # a loader that HAS the method but cannot shard what it wraps (DeviceLoader over a custom sample source without
                # shard(rank, world)): the fingerprinted stride mode below still works for it
                self.mode, self.inner = "stride", None
        if self.inner is None:
            try:
                from torch.utils.data import DataLoader
            except Exception:                       # pragma: no cover
                DataLoader = ()
            if DataLoader and isinstance(loader, DataLoader) and loader.batch_size is not None and hasattr(loader.dataset, "__len__"):
                self.sampler = EpochShardSampler(len(loader.dataset), rank, world, loader.batch_size, shuffle=_is_shuffling(loader), seed=seed)
                # everything else the caller configured travels along.  Replaced on purpose: the sampler (and with it a
                # RandomSampler's own generator / seed -- the permutation must be the SAME on every rank, so it is seeded seed + epoch here)
                # and drop_last (the ranks' counts must agree)
                extra = {}
                if loader.num_workers > 0 and getattr(loader, "prefetch_factor", None) is not None:
                    extra["prefetch_factor"] = loader.prefetch_factor
                for k in ("timeout", "generator", "multiprocessing_context", "pin_memory_device"):
                    v = getattr(loader, k, None)
                    if v not in (None, "", 0) and not (k == "multiprocessing_context" and loader.num_workers == 0):
                        extra[k] = v
                self.inner = DataLoader(loader.dataset, batch_size=loader.batch_size, sampler=self.sampler, num_workers=loader.num_workers,
                                        collate_fn=loader.collate_fn, pin_memory=loader.pin_memory, drop_last=True,
                                        worker_init_fn=loader.worker_init_fn, persistent_workers=getattr(loader, "persistent_workers", False), **extra)
                self.mode = "sampler"

    def __len__(self):
        if self.inner is not None:
            return len(self.inner)
        return len(self.loader) // self.world

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def _agreed(self, mine):
        """the number of batches EVERY rank will train on this epoch: the minimum over the ranks of what each one has (one small collective
        on the host group per epoch).  Loaders whose lengths disagree between ranks (a file missing on one host, a dataset of another
        size) would otherwise leave the better-supplied ranks waiting in a gradient all-reduce that the others never enter."""
        if mine is None or not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return mine
        v = torch.tensor([int(mine)], dtype=torch.int64)
        dist.all_reduce(v, op=dist.ReduceOp.MIN, group=_host_group(self.group))
        return int(v[0])

    def _bounded(self, it, n):
        out, it = 0, iter(it)
        while n is None or out < n:                 # checked BEFORE drawing: nothing is pulled (decoded, copied) past the agreed count
            try:
                b = next(it)
            except StopIteration:
                break
            yield b
            out += 1
        if n is not None and out < n:
            raise RuntimeError("ShardedLoader: rank %d ran out of batches after %d of the %d the ranks agreed on for this epoch -- the "
                               "other ranks are about to wait for its gradients; aborting this rank so that the launcher stops the run" % (self.rank, out, n))

    def __iter__(self):
        epoch, self.epoch = self.epoch, self.epoch + 1
        if self.mode == "index":
            yield from self._bounded(self.inner, self._agreed(len(self.inner) if hasattr(self.inner, "__len__") else None))
            return
        if self.mode == "sampler":
            self.sampler.set_epoch(epoch)
            yield from self._bounded(self.inner, self._agreed(len(self.inner)))
            return
        n = self._agreed(len(self) if hasattr(self.loader, "__len__") else None)
        if n == 0:
            return                                  # agreed by all ranks: nobody enters the fingerprint collective
        it = iter(self.loader)
        try:
            first = next(it)
        except StopIteration:
            first = _NO_BATCH
        # EVERY rank enters this collective exactly once per epoch, with a sentinel when its loader is empty (a rank that
        # drew nothing used to skip it and leave the others waiting, or pair up with their next collective)
        _assert_same_first_batch(first, self.group)
        if first is _NO_BATCH:
            if n:
                raise RuntimeError("ShardedLoader: rank %d has no batches but the ranks agreed on %d for this epoch" % (self.rank, n))
            return
        import itertools
        pending, out = [], 0
        for b in itertools.chain([first], it):
            pending.append(b)
            if len(pending) == self.world:          # a full round: hand out this rank's batch
                yield pending[self.rank]
                pending, out = [], out + 1
                if n is not None and out >= n:
                    return


def _is_shuffling(loader):
    try:
        from torch.utils.data import RandomSampler
        return isinstance(loader.sampler, RandomSampler)
    except Exception:                               # pragma: no cover
        return False


class EpochShardSampler:
    """indices of rank r for one epoch: the (optionally shuffled, epoch-seeded -- identical on every rank) permutation of range(n), cut to a
    multiple of world * batch_size, dealt out batch-wise: global batch g goes to rank g % world.  torch's DistributedSampler deals out single
    indices; dealing out whole batches keeps a rank's batches equal to the ones the single-process loader would have formed."""

    def __init__(self, n, rank, world, batch_size, shuffle=True, seed=0):
        self.n, self.rank, self.world, self.bs, self.shuffle, self.seed, self.epoch = n, rank, world, batch_size, shuffle, seed, 0
        self.rounds = n // (world * batch_size)

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __len__(self):
        return self.rounds * self.bs

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        else:
            order = list(range(self.n))
        for r in range(self.rounds):
            base = (r * self.world + self.rank) * self.bs
            yield from order[base:base + self.bs]


def _fingerprint(obj):
    import hashlib
    h = hashlib.sha256()

    def feed(o):
        if isinstance(o, dict):
            for k in sorted(o):
                h.update(str(k).encode())
                feed(o[k])
        elif isinstance(o, (list, tuple)):
            for v in o:
                feed(v)
        elif torch.is_tensor(o):
            t = o.detach()
            h.update(str(tuple(t.shape)).encode())
            flat = t.reshape(-1)
            step = max(1, flat.numel() // 4096)
            h.update(flat[::step].cpu().contiguous().numpy().tobytes())
        elif hasattr(o, "tobytes"):
            h.update(o.tobytes()[:1 << 16])
        else:
            h.update(repr(o).encode())
    feed(obj)
    return int.from_bytes(h.digest()[:7], "little")


_NO_BATCH = object()          # what a rank whose loader is empty brings to the fingerprint collective


def _assert_same_first_batch(batch, group=None):
    """fallback sharding only: every rank must draw the same batch order -- compare a fingerprint of the epoch's first batch over the host group"""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    fp = 0 if batch is _NO_BATCH else max(1, _fingerprint(batch))
    v = torch.tensor([fp, -fp], dtype=torch.int64)
    dist.all_reduce(v, op=dist.ReduceOp.MAX, group=_host_group(group))
    if int(v[0]) != -int(v[1]):
        raise RuntimeError("ShardedLoader: the ranks drew different first batches from a loader that is sharded by iteration (it has no "
                           "shard(rank, world) and is not a torch DataLoader): their shards would overlap.  Seed the loader identically on "
                           "every rank or give it a shard(rank, world) method")
