"""CPU (gloo, world_size 2 where a collective is involved): how parallel.ShardedLoader deals a loader's batches out to the ranks of a
data-parallel run (the reference trainer is single-process, training/train.py:145-191).  Round 5 / ADVICE r4: sharding by
INDEX (a rank pays the input pipeline for its own batches only), an epoch-seeded shared permutation for torch DataLoaders, a fingerprint check on
the iterate-everything fallback, and an agreed batch count per epoch so that no rank is left waiting in a collective."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Counting(torch.utils.data.Dataset):
    """dataset that records which items were actually produced"""

    def __init__(self, n):
        self.n, self.touched = n, []

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        self.touched.append(i)
        return {"image": torch.full((3, 2, 2), float(i)), "idx": torch.tensor(i)}


def test_index_level_shard_of_the_synthetic_loaders_touches_only_the_ranks_own_batches():
    from footprints_amd.datasets.device_path import SyntheticSampleSource
    from footprints_amd.parallel import ShardedLoader
    src = SyntheticSampleSource(2, 4, 6, steps=7, pool=14)
    whole = [[id(s) for s in b] for b in src]
    for world in (2, 3):
        seen = []
        for r in range(world):
            sh = ShardedLoader(src, r, world)
            assert sh.mode == "index" and len(sh) == 7 // world
            got = [[id(s) for s in b] for b in sh]
            assert got == whole[r:(7 // world) * world:world]
            seen += got
        assert len(seen) == (7 // world) * world


def test_torch_dataloader_is_rebuilt_over_a_rank_strided_epoch_seeded_sampler():
    from torch.utils.data import DataLoader
    from footprints_amd.parallel import ShardedLoader
    n, bs, world = 26, 3, 2
    for shuffle in (False, True):
        per_epoch = []
        for epoch in range(2):
            got = {}
            for r in range(world):
                ds = _Counting(n)
                sh = ShardedLoader(DataLoader(ds, batch_size=bs, shuffle=shuffle), r, world, seed=5)
                assert sh.mode == "sampler" and len(sh) == n // (bs * world)
                sh.set_epoch(epoch)
                got[r] = [b["idx"].tolist() for b in sh]
                assert sorted(ds.touched) == sorted(i for b in got[r] for i in b)        # the rank produced ONLY its own items
                assert all(len(b) == bs for b in got[r])
            flat = [i for r in range(world) for b in got[r] for i in b]
            assert len(flat) == len(set(flat)) == (n // (bs * world)) * bs * world       # disjoint, equal counts
            if not shuffle:                                                              # rank r = global batches r, r + world, ...
                assert got[0][0] == [0, 1, 2] and got[1][0] == [3, 4, 5] and got[0][1] == [6, 7, 8]
            per_epoch.append(got)
        if shuffle:
            assert per_epoch[0] != per_epoch[1]                                          # another permutation every epoch ...
        else:
            assert per_epoch[0] == per_epoch[1]


class _Plain:
    """no shard(), not a DataLoader: the iterate-everything fallback"""

    def __init__(self, tags):
        self.tags, self.dataset = list(tags), range(len(tags))

    def __len__(self):
        return len(self.tags)

    def __iter__(self):
        for t in self.tags:
            yield {"image": torch.full((1, 3, 2, 2), float(t))}


def _worker(rank, world, port, case, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from footprints_amd.parallel import DistContext
    ctx = DistContext.from_env(use_cuda=False)
    try:
        if case == "same_order":
            sh = ctx.shard(_Plain(range(7)))
            q.put((rank, "ok", [int(b["image"].flatten()[0]) for b in sh]))
        elif case == "different_order":                         # rank 1 shuffled differently: overlapping shards -> must raise on every rank
            sh = ctx.shard(_Plain(range(7) if rank == 0 else [3, 1, 2, 0, 4, 5, 6]))
            try:
                list(sh)
                q.put((rank, "no error", None))
            except RuntimeError as e:
                q.put((rank, "raised", str(e)[:60]))
        elif case == "unequal_counts":                          # rank 1's loader is two global batches short: both train on the common count
            from footprints_amd.training.train import SyntheticLoader

            class L(SyntheticLoader):                           # index-sharded loader without device batches
                def __init__(self, steps):
                    self.steps, self.first, self.stride, self.batches, self.dataset = steps, 0, 1, [{"image": torch.zeros(1)}], range(steps)
            sh = ctx.shard(L(10 if rank == 0 else 6))
            q.put((rank, "ok", len(list(sh))))
        elif case == "runs_dry":                                # rank 1's loader promises 5 batches and delivers 3: it must abort loudly
            from footprints_amd.training.train import SyntheticLoader

            class Lying(SyntheticLoader):
                def __init__(self, steps, real):
                    self.steps, self.first, self.stride, self.batches, self.dataset, self.real = steps, 0, 1, [{"image": torch.zeros(1)}], range(steps), real

                def shard(self, rank_, world_):
                    v = super().shard(rank_, world_)
                    v.real = self.real
                    return v

                def __iter__(self):
                    for i, b in enumerate(super().__iter__()):
                        if i >= self.real:
                            return
                        yield b
            sh = ctx.shard(Lying(10, 5 if rank == 0 else 3))
            try:
                n = len(list(sh))
                q.put((rank, "ok", n))
            except RuntimeError as e:
                q.put((rank, "raised", str(e)[:40]))
        elif case == "one_rank_empty":                          # no __len__, rank 1 yields nothing -- every rank must still enter the
            class NoLen:                                        # fingerprint collective (sentinel) and both must raise, nobody hangs
                def __init__(self, tags):
                    self.tags = list(tags)

                def __iter__(self):
                    for t in self.tags:
                        yield {"image": torch.full((1, 3, 2, 2), float(t))}
            sh = ctx.shard(NoLen(range(4) if rank == 0 else []))
            try:
                list(sh)
                q.put((rank, "no error", None))
            except RuntimeError as e:
                q.put((rank, "raised", str(e)[:40]))
        elif case == "zero_agreed":                             # a loader with fewer batches than ranks: 0 rounds agreed, nothing drawn, no collective left open
            sh = ctx.shard(_Plain(range(1)))
            q.put((rank, "ok", len(list(sh))))
            assert ctx.mean_losses({"loss": float(rank)})["loss"] == 0.5     # the NEXT collective pairs up correctly on both ranks
        elif case == "shard_method_cannot_shard":               # a loader WITH shard() that raises TypeError falls back to the stride mode
            class Wrapped(_Plain):
                def shard(self, r, w):
                    raise TypeError("the sample source has no shard(rank, world)")
            sh = ctx.shard(Wrapped(range(7)))
            q.put((rank, sh.mode, [int(b["image"].flatten()[0]) for b in sh]))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _run(case):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r = q.get(timeout=300)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_fallback_sharding_checks_that_the_ranks_draw_the_same_order():
    res = _run("same_order")
    assert res[0] == ("ok", [0, 2, 4]) and res[1] == ("ok", [1, 3, 5])
    res = _run("different_order")
    assert res[0][0] == "raised" and res[1][0] == "raised", res


def test_ranks_agree_on_the_batch_count_of_an_epoch():
    """a rank whose loader is shorter must not leave the other one waiting in a gradient all-reduce: both iterate the minimum"""
    res = _run("unequal_counts")
    assert res[0] == ("ok", 3) and res[1] == ("ok", 3), res


def test_a_rank_that_runs_dry_below_the_agreed_count_aborts_instead_of_stranding_the_others():
    """every rank agreed on 5 batches for the epoch; rank 1's loader ends after 3: it raises (the launcher then stops the run) rather than
    returning from the epoch while rank 0 walks into the next gradient all-reduce alone"""
    res = _run("runs_dry")
    assert res[0] == ("ok", 5), res
    assert res[1][0] == "raised" and "ran out of batches" in res[1][1], res


def test_every_rank_enters_the_fingerprint_collective_even_with_an_empty_loader():
    res = _run("one_rank_empty")
    assert res[0][0] == "raised" and res[1][0] == "raised", res
    res = _run("zero_agreed")
    assert res[0] == ("ok", 0) and res[1] == ("ok", 0), res


def test_a_loader_whose_shard_method_cannot_shard_falls_back_to_the_stride_mode():
    res = _run("shard_method_cannot_shard")
    assert res[0] == ("stride", [0, 2, 4]) and res[1] == ("stride", [1, 3, 5]), res


def test_device_loader_over_a_custom_source_still_shards_by_stride():
    """DeviceLoader always has a shard attribute; over a sample source without shard(rank, world) ShardedLoader used to fail at construction"""
    from footprints_amd.datasets.device_path import DeviceLoader
    from footprints_amd.parallel import ShardedLoader
    dl = DeviceLoader([[("img", {})]] * 6, assembler=None)
    sh = ShardedLoader(dl, 0, 2)
    assert sh.mode == "stride" and len(sh) == 3


def test_bounded_iteration_draws_nothing_past_the_agreed_count():
    from footprints_amd.parallel import ShardedLoader
    drawn = []

    def gen():
        for i in range(10):
            drawn.append(i)
            yield i
    sh = ShardedLoader(_Plain(range(2)), 0, 1)
    assert list(sh._bounded(gen(), 3)) == [0, 1, 2] and drawn == [0, 1, 2]
