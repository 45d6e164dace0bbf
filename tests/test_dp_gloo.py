"""CPU, world_size 2, gloo: the data-parallel gradient exchange (footprints_amd/parallel.py) against the
N-shard oracle: rank-summed buckets x 1/world == mean of per-shard gradients with
per-shard BatchNorm statistics, and both ranks end up with identical buffers."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _layout():
    from footprints_amd import FootprintNetwork
    m = FootprintNetwork(pretrained=False)
    names, offs, total = [], [], 0
    for n, p in m.live_named_parameters():
        names.append(n)
        offs.append(total)
        total += (p.numel() + 3) // 4 * 4
    return names, offs, total


def _shard_grads(rank):
    from oracle import restatement as R
    torch.set_num_threads(2)
    P, B = R.make_state(tag="dp")
    tr = R.OracleTrainer(P, B)
    tr.forward_backward(R.make_batch(1, 64, 64, tag="dp.shard%d" % rank))
    return {k: p.grad for k, p in tr.P.items() if p.grad is not None}


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from footprints_amd.parallel import GradReducer
    names, offs, total = _layout()
    g = _shard_grads(rank)
    flat = torch.zeros(total)
    for n, o in zip(names, offs):
        flat[o:o + g[n].numel()] = g[n].flatten()
    red = GradReducer(flat, names, offs, max_elems=2 << 20)
    assert red.world == 2 and abs(red.grad_scale - 0.5) < 1e-12
    for stage in ("mask_decoder", "depth_decoder", "encoder.layer4", "encoder.layer3"):   # reported by the schedule
        red.stage_ready(stage)
    red.finish()                                                                           # flushes the rest
    q.put((rank, (flat * red.grad_scale).numpy()))        # by value: a shared-memory tensor handle can outlive its sender (flaky ConnectionResetError)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_mean_matches_shard_oracle():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: torch.from_numpy(a) for r, a in (q.get(timeout=300) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.equal(res[0], res[1])                       # identical on every rank => identical weights after Adam
    names, offs, total = _layout()
    g0, g1 = _shard_grads(0), _shard_grads(1)
    for n, o in zip(names, offs):
        ref = 0.5 * (g0[n] + g1[n]).flatten()
        got = res[0][o:o + ref.numel()]
        assert torch.allclose(got, ref, rtol=1e-6, atol=1e-9), n


def _comm_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from footprints_amd import parallel
    comm = parallel.get_communicator(None)                 # no GPU here: loading RCCL may work, creating a communicator cannot
    q.put((rank, comm is None, parallel.get_communicator(None, create=False) is None))
    dist.barrier()
    dist.destroy_process_group()


def test_communicator_bootstrap_failure_is_agreed_upon_and_nobody_hangs():
    """fp_comm bootstrap on a box without GPUs, two ranks: rank 0's unique id travels (or its failure does), every rank's
    ncclCommInitRank fails, the ranks agree on the torch.distributed fallback through the MIN all-reduce -- no rank is left waiting in a
    collective the other never enters (the pre-flight / broadcast-the-failure logic of footprints_amd/parallel.py)"""
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without GPUs (with GPUs the communicator is created: tests/test_gpu_dp.py)")
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_comm_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == [(0, True, True), (1, True, True)]
