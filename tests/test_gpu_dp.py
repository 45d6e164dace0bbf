"""Data-parallel training step on the GPU path, world size 2.

RCCL refuses two ranks on one device, and the GPU box of the test tier has a single GPU, so the two ranks share cuda:0 and
exchange gradients over gloo (it stages CUDA tensors through the host): same GradReducer / TrainStep / fused-Adam code as under
"nccl", only the transport differs.  Checked: (a) both ranks hold bit-identical weights after every step, (b) they equal a
single-process emulation that sums the two shards' gradients with the drop-in autograd path and applies Adam with grad_scale 1/2.
Infrastructure failures (no free port, spawn trouble) skip; numerical mismatches fail.
"""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

STEPS, B, H, W = 5, 2, 64, 96      # 2 eager + 1 recorded + 2 replayed steps: the launch plan in pieces around the bucket all-reduces


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _state(tag="dp"):
    from oracle import restatement as R
    return R.make_state(tag=tag)


def _load(model, P, Bf):
    sd = model.state_dict()
    model.load_state_dict({k: (P[k] if k in P else Bf[k]).to(sd[k].dtype) for k in sd})
    return model


def _shard(rank):
    from oracle import restatement as R
    return {k: v.cuda() for k, v in R.make_batch(B, H, W, tag="dp.shard%d" % rank).items()}


class _GlooBackedComm:
    """stands in for footprints_amd.parallel.Communicator where RCCL cannot run (two ranks on one GPU): same interface, the collective
    itself over the gloo group after draining the stream it was issued on -- so that GradReducer's "rccl" branch (library events
    between the gradient-writing streams and the communication stream, all-reduces on the mask decoder's weight-gradient stream,
    the compute stream joined before Adam) runs with a real world of two"""

    def __init__(self, dist):
        self.dist, self.rank, self.world, self.handle = dist, dist.get_rank(), dist.get_world_size(), None

    def allreduce(self, t, stream):
        stream.synchronize()
        self.dist.all_reduce(t)

    def broadcast(self, t, root, stream):
        stream.synchronize()
        self.dist.broadcast(t, src=root)

    def destroy(self):
        pass


def _worker(rank, world, port, q, overlap, fake_rccl=False):
    try:
        import torch.distributed as dist
        os.environ["FP_DP_OVERLAP"] = "1" if overlap else "0"     # read when footprints_amd.parallel is imported
        os.environ["FP_DP_ALLOW_SHARED_GPU"] = "1"                # two ranks on the test box's single GPU: has to be asked for (round 5)
        if fake_rccl:
            os.environ["FP_DP_TRANSPORT"] = "rccl"
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from footprints_amd import parallel
        from footprints_amd.model_manager import ModelManager
        from footprints_amd.parallel import broadcast_state
        from footprints_amd.training.train import TrainStep
        if fake_rccl:
            fake = _GlooBackedComm(dist)
            parallel.get_communicator = lambda group=None, create=True: fake
        mm = ModelManager()
        P, Bf = _state("dp" if rank == 0 else "dp.other")     # rank 1 starts from different weights: broadcast_state must fix that
        _load(mm.model, P, Bf)
        mm.model.eval()
        with torch.no_grad():                                   # a forward BEFORE the broadcast: the engine now holds packed / BN-folded copies
            mm.model(_shard(rank)["image"])                     # of this rank's own weights, which broadcast_state has to invalidate
        mm.model.train()
        broadcast_state(mm.model)
        ts = TrainStep(mm.model, mm.optimiser, distributed=True, plan=False if fake_rccl else None)     # the stand-in is not recordable
        assert ts.reducer is not None and ts.reducer.world == 2 and ts.reducer.overlap == bool(overlap)
        assert ts.reducer.transport == ("rccl" if fake_rccl else "torch")
        if fake_rccl:
            assert ts.reducer.stream.cuda_stream == ts.eng.dwg[0].cuda_stream        # the all-reduces ride an engine stream, not a sixth one
        batch = _shard(rank)
        losses = []
        for _ in range(STEPS):
            losses.append(float(ts(batch)[20]))
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().flatten() for p in mm.model.parameters()]).cpu().numpy()
        q.put((rank, "ok", flat, losses))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:                                      # report instead of dying silently
        import traceback
        q.put((rank, "error", traceback.format_exc(), repr(e)))


# buckets reduced after backward (FP_DP_OVERLAP=0) / as soon as each stage is complete (default); and the reducer's "rccl" branch with a
# gloo-backed stand-in for the communicator (world of two)
@pytest.mark.parametrize("overlap,fake_rccl", [(False, False), (True, False), (True, True), (False, True)])
def test_two_ranks_share_weights_and_match_summed_shard_gradients(overlap, fake_rccl):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    try:
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, q, overlap, fake_rccl)) for r in range(2)]
        for p in procs:
            p.start()
    except OSError as e:
        pytest.skip("cannot spawn the two ranks: %r" % (e,))
    res = {}
    try:
        for _ in range(2):
            r = q.get(timeout=600)
            res[r[0]] = r
    except Exception:
        for p in procs:
            if p.is_alive():
                p.terminate()
        pytest.skip("the two ranks did not report within 600 s (rendezvous / transport problem on this box)")
    for p in procs:
        p.join(timeout=120)
    for r in res.values():
        if r[1] == "error":
            if "Connection" in r[2] or "Address already in use" in r[2] or "timed out" in r[2].lower():
                pytest.skip("gloo rendezvous failed: " + r[3])
            if "ProcessGroupGloo" in r[2] and "CUDA" in r[2]:
                pytest.skip("this gloo build cannot move CUDA tensors: " + r[3])
            raise AssertionError("rank %d failed:\n%s" % (r[0], r[2]))
    w0, w1 = res[0][2], res[1][2]
    assert np.array_equal(w0, w1), "ranks diverged"                       # (a)

    # (b) single-process emulation: gradients of shard 0 + shard 1 through the drop-in autograd path, Adam with grad_scale 1/2
    from footprints_amd.model_manager import ModelManager
    from footprints_amd.training.losses import LossManager
    mm = ModelManager()
    P, Bf = _state("dp")
    _load(mm.model, P, Bf)
    mm.optimiser.grad_scale = 0.5
    lm = LossManager((0.1, 100), 0.25, compute_viz=False)
    shards = [_shard(0), _shard(1)]
    for _ in range(STEPS):
        mm.model.train()
        mm.model.zero_grad()
        for sh in shards:
            lm(mm.model(sh["image"]), sh)["loss"].backward()
        mm.optimiser.step()
    ref = torch.cat([p.detach().flatten() for p in mm.model.parameters()]).cpu().numpy()
    err = np.abs(w0.astype(np.float64) - ref).max() / np.abs(ref).max()
    assert err < 1e-6, err
    assert res[0][3][-1] != res[1][3][-1]                                 # different shards => different per-rank losses


def _nccl_worker(port, q, transport):
    try:
        import torch.distributed as dist
        os.environ["FP_DP_FORCE"] = "1"                           # issue the bucket collectives although world == 1
        os.environ["FP_DP_TRANSPORT"] = transport
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(0)
        if transport == "torch":                                  # framework collectives need the framework's group; fp_comm_* does not
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        from footprints_amd.model_manager import ModelManager
        from footprints_amd.training.train import TrainStep
        P, Bf = _state("dp")
        batch = _shard(0)
        res = []
        for mode in ("single", "after", "overlap"):
            mm = ModelManager()
            _load(mm.model, P, Bf)
            ts = TrainStep(mm.model, mm.optimiser, distributed=(mode != "single"))
            if ts.reducer is not None:
                ts.reducer.overlap = (mode == "overlap")
                assert ts.reducer.force and ts.reducer.world == 1 and len(ts.reducer.buckets) >= 7
                assert ts.reducer.transport == transport and ts.reducer.plan_recordable == (transport == "rccl")
            losses = [float(ts(batch)[20]) for _ in range(6)]        # 2 eager + 1 recorded + 3 replayed (torch: in pieces around the bucket all-reduces;
                                                                     # rccl: in one piece, the all-reduces are nodes of the plan)
            assert len(ts._plans) == 1
            torch.cuda.synchronize()
            res.append((losses, torch.cat([p.detach().flatten() for p in mm.model.parameters()]).cpu().numpy()))
        q.put(("ok", res))
        from footprints_amd.parallel import destroy_communicators
        destroy_communicators()
        if dist.is_initialized():
            dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put(("error", traceback.format_exc(), repr(e)))


@pytest.mark.parametrize("transport", ["rccl", "torch"])
def test_rccl_path_world_of_one_executes_and_is_exact(transport):
    """The RCCL code path executed on the 1-GPU box with a world of one rank, through both transports of GradReducer: "rccl" = the
    library's own fp_comm_* entry points (ncclCommInitRank / ncclAllReduce through csrc/comm.cpp, the all-reduces recorded into the
    launch plan), "torch" = torch.distributed collectives on an "nccl" group (the fallback).  Bucket all-reduces on the comm stream,
    event gating, both schedules: a sum over one rank is the identity, so the steps must be BIT-identical to the non-distributed
    TrainStep -- a wrong wait / a bucket reduced before its gradients landed would show.  Steps 4-6 replay the recorded launch plan."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), q, transport))
    p.start()
    try:
        r = q.get(timeout=600)
    except Exception:
        if p.is_alive():
            p.terminate()
        pytest.skip("the nccl worker did not report within 600 s")
    p.join(timeout=120)
    if r[0] == "error":
        if "Address already in use" in r[1] or "Connection" in r[1]:
            pytest.skip("rendezvous failed: " + r[2])
        raise AssertionError("nccl worker failed:\n" + r[1])
    (l0, w0), (l1, w1), (l2, w2) = r[1]
    assert l0 == l1 == l2
    assert np.array_equal(w0, w1) and np.array_equal(w0, w2)


def test_engine_invalidate_after_data_write():
    """`p.data.copy_()` does not move Parameter._version: without Engine.invalidate() the convolutions would keep the packed
    copies of the OLD weights.  forward -> overwrite all weights -> invalidate -> forward == fresh model."""
    from footprints_amd import FootprintNetwork
    Pa, Ba = _state("dp")
    Pb, Bb = _state("dp.other")
    img = _shard(0)["image"]
    m = _load(FootprintNetwork(pretrained=False), Pa, Ba).cuda().eval()
    fresh = _load(FootprintNetwork(pretrained=False), Pb, Bb).cuda().eval()
    with torch.no_grad():
        m(img)
        sd = m.state_dict()
        for k, v in {**Pb, **Bb}.items():
            sd[k].data.copy_(v)
        m.engine().invalidate()
        a, b = m(img), fresh(img)
    for k in a:
        assert torch.equal(a[k], b[k]), k


# ---- one rank per device over RCCL: runs by itself wherever pytest sees two or more GPUs ------------------------------------------------
def _rccl_worker(rank, world, port, q):
    try:
        import torch.distributed as dist
        os.environ.pop("FP_DP_TRANSPORT", None)                   # the default choice must be rccl when every rank has a device
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(rank)
        dist.init_process_group("gloo", rank=rank, world_size=world)      # host plumbing only: rendezvous, unique id, barriers
        from footprints_amd.model_manager import ModelManager
        from footprints_amd.parallel import broadcast_state, destroy_communicators, get_communicator
        from footprints_amd.training.train import TrainStep
        mm = ModelManager()
        P, Bf = _state("dp" if rank == 0 else "dp.other")
        _load(mm.model, P, Bf)
        broadcast_state(mm.model)
        ts = TrainStep(mm.model, mm.optimiser, distributed=True)
        comm = get_communicator(None, create=False)
        info = {"transport": ts.reducer.transport, "comm_world": comm.world if comm is not None else 0, "reducer_world": ts.reducer.world,
                "recordable": ts.reducer.plan_recordable, "device": torch.cuda.current_device()}
        batch = _shard(rank)
        losses = [float(ts(batch)[20]) for _ in range(STEPS)]     # 2 eager + record + 2 replays, the all-reduces inside the plan
        info["plans"] = len(ts._plans)
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().flatten() for p in mm.model.parameters()]).cpu().numpy()
        q.put((rank, "ok", flat, losses, info))
        dist.barrier()
        destroy_communicators()
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, "error", traceback.format_exc(), repr(e), {}))


def test_one_rank_per_device_over_rccl_matches_the_n_shard_oracle():
    """SURVEY.md section 8(e) / north_star config #4 at whatever size the box offers: N = min(device_count, 8) ranks, one per GPU, gradients
    over the library's own RCCL communicator (fp_comm_*).  Bit-identical weights on every rank after 5 steps (2 eager + the recorded launch
    plan + 2 replays), equal to the single-process N-shard emulation (sum of the shards' gradients, Adam with grad_scale 1/N) at 1e-6, the
    transport really is "rccl" and the communicator really spans N ranks.  The reference has no counterpart (README.md:128: one GPU)."""
    n = min(torch.cuda.device_count(), 8)
    if n < 2:
        pytest.skip("ONE GPU visible: RCCL with more than one rank cannot run here (RCCL refuses two ranks per device); this case runs "
                    "by itself on any box with >= 2 GPUs")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, n, port, q)) for r in range(n)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(n):
            r = q.get(timeout=900)
            res[r[0]] = r
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.terminate()
    for r in res.values():
        assert r[1] == "ok", "rank %d failed:\n%s" % (r[0], r[2])
    for r in range(n):
        info = res[r][4]
        assert info["transport"] == "rccl" and info["recordable"] and info["comm_world"] == n and info["reducer_world"] == n, info
        assert info["device"] == r and info["plans"] == 1, info
        assert np.array_equal(res[0][2], res[r][2]), "rank %d diverged from rank 0" % r
    from footprints_amd.model_manager import ModelManager
    from footprints_amd.training.losses import LossManager
    mm = ModelManager()
    P, Bf = _state("dp")
    _load(mm.model, P, Bf)
    mm.optimiser.grad_scale = 1.0 / n
    lm = LossManager((0.1, 100), 0.25, compute_viz=False)
    shards = [_shard(r) for r in range(n)]
    for _ in range(STEPS):
        mm.model.train()
        mm.model.zero_grad()
        for sh in shards:
            lm(mm.model(sh["image"]), sh)["loss"].backward()
        mm.optimiser.step()
    ref = torch.cat([p.detach().flatten() for p in mm.model.parameters()]).cpu().numpy()
    err = np.abs(res[0][2].astype(np.float64) - ref).max() / np.abs(ref).max()
    assert err < 1e-6, err


# ---- the trainer itself as one replica of two (ranks share the test box's GPU: gloo transport) -------------------------------------------
def _trainer_worker(rank, world, port, folder, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK="0",
                          FP_DP_ALLOW_SHARED_GPU="1")
        torch.cuda.set_device(0)
        from footprints_amd.main import main
        from footprints_amd.training import train as T
        made = []
        orig = T.TrainManager.__init__

        def spy(self, *a, **k):
            orig(self, *a, **k)
            made.append(self)
        T.TrainManager.__init__ = spy
        main(["--mode", "train", "--synthetic_steps", "8", "--epochs", "1", "--batch_size", "2", "--height", "64", "--width", "96",
              "--val_batches", "1", "--log_path", folder, "--model_name", "dp"])
        tm = made[0]
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().flatten() for p in tm.model.parameters()]).cpu().numpy()
        q.put((rank, "ok", flat, tm.step, tm.train_step.reducer is not None and tm.train_step.reducer.world, tm.history))
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    except BaseException as e:
        import traceback
        q.put((rank, "error", traceback.format_exc(), repr(e), None, None))


def test_main_trains_data_parallel_under_a_launcher_environment(tmp_path):
    """`python -m footprints_amd.main --mode train ...` under WORLD_SIZE = 2 (what torch.distributed.run sets): each rank trains on every
    second batch, weights stay bit-identical, rank 0 alone writes the epoch checkpoint, both hold the same rank-averaged loss history."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(2):
            r = q.get(timeout=600)
            res[r[0]] = r
    except Exception:
        for p in procs:
            if p.is_alive():
                p.terminate()
        pytest.skip("the two ranks did not report within 600 s (rendezvous / transport problem on this box)")
    for p in procs:
        p.join(timeout=120)
    for r in res.values():
        if r[1] == "error":
            if "Connection" in r[2] or "Address already in use" in r[2]:
                pytest.skip("gloo rendezvous failed: " + r[3])
            raise AssertionError("rank %d failed:\n%s" % (r[0], r[2]))
    assert np.array_equal(res[0][2], res[1][2]), "ranks diverged"
    assert res[0][3] == res[1][3] == 4 and res[0][4] == res[1][4] == 2          # 8 global batches -> 4 steps per rank; reducer world 2
    assert res[0][5]["train"] == res[1][5]["train"] and res[0][5]["val"] == res[1][5]["val"]
    models = os.path.join(str(tmp_path), "dp", "models")
    assert sorted(os.listdir(models)) == ["weights_0"] and sorted(os.listdir(os.path.join(models, "weights_0"))) == ["model.pth", "optimiser.pth"]


# ---- ranks that share a physical GPU without having asked for it: an error, not a silent downgrade to gloo ---------------------
def _shared_worker(rank, world, port, q):
    try:
        import torch.distributed as dist
        os.environ.pop("FP_DP_ALLOW_SHARED_GPU", None)
        os.environ.pop("FP_DP_TRANSPORT", None)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from footprints_amd import parallel
        ident = parallel._physical_device_id()
        try:
            parallel._pick_transport(torch.zeros(4, device="cuda"), None, False)
            q.put((rank, "no error", ident))
        except RuntimeError as e:
            q.put((rank, "raised", ident, str(e)[:80]))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException as e:
        import traceback
        q.put((rank, "error", traceback.format_exc()))


def test_ranks_on_one_physical_gpu_are_refused_unless_asked_for():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shared_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r = q.get(timeout=300)
        res[r[0]] = r
    for p in procs:
        p.join(timeout=60)
    assert res[0][1] == res[1][1] == "raised", res
    assert res[0][2] == res[1][2] and not res[0][2].startswith("logical:"), res        # the same PHYSICAL identity (uuid / PCI address) on both ranks


# ---- bench.py's forced data-parallel line (world of one): the RCCL communicator counts its ranks itself, and the exchange costs nothing measurable --
def test_bench_forced_data_parallel_line_counts_its_ranks_and_prices_the_exchange():
    """`bench.py --force-dist` (what the N > 1 launches run, in a world of one): transport rccl, ncclCommCount == 1 (bench.py exits non-zero when the
    communicator's own count differs from the ranks it was launched with), the bucket all-reduces inside the recorded plan, and
    `exposed_communication` -- the same step with and without its all-reduces, alternating blocks in one process -- within 5 % of the step."""
    import json
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("FP_DP_TRANSPORT", "FP_DP_FORCE", "WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "10", "--warmup", "5", "--sustain", "0",
                        "--no-cpu-baseline", "--no-loader", "--no-other-format", "--no-kernel-events"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.stderr or r.stdout)[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    assert len(line) < 4096                                     # round 6: the stdout line is the compact record ...
    d = json.loads(line)
    gx = d["config"]["gradient_exchange"]
    assert d["rccl_ranks"] == 1 and gx["transport"] == "rccl" and gx["in_launch_plan"] and gx["buckets"] >= 7
    assert abs(gx["exposed_ms"]) <= 0.05 * gx["step_ms_without_exchange"], gx
    with open(os.path.join(ROOT, d["detail"])) as fh:           # ... and the full record sits beside it
        full = json.load(fh)
    assert full["value"] == d["value"] and full["config"]["gradient_exchange"]["exposed_communication"]["exposed_ms"] == gx["exposed_ms"]
