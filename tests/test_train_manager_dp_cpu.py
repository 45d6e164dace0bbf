"""CPU, world_size 2, gloo: the data-parallel HOST logic of TrainManager (footprints_amd/training/train.py, parallel.DistContext) -- per-rank
shard of the loader, rank-0-only console line and checkpoint, the logged losses averaged over the ranks.  There is no CPU compute path in
the product, so the step and the model manager are stand-ins; the loop around them is the real one (reference loop:
footprints/training/train.py:145-191, single process)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeScheduler:
    def __init__(self):
        self.steps = 0

    def get_last_lr(self):
        return [1e-4]

    def step(self):
        self.steps += 1


class _FakeModelManager:
    def __init__(self, folder):
        self.save_folder = folder
        self.model = torch.nn.Linear(1, 1)
        self.optimiser, self.scheduler = None, _FakeScheduler()
        self.saved = []

    def save_model(self, folder_name):
        os.makedirs(os.path.join(self.save_folder, folder_name), exist_ok=True)
        open(os.path.join(self.save_folder, folder_name, "model.pth"), "w").write("rank-0")
        self.saved.append(folder_name)


class _FakeStep:
    """21 'losses' = the batch's tag: lets the test see which batches a rank trained on"""

    def __init__(self):
        self.seen = []

    def __call__(self, batch):
        tag = float(batch["image"].flatten()[0])
        self.seen.append(tag)
        return torch.full((21,), tag)


class _TaggedLoader:
    def __init__(self, n):
        self.n, self.dataset = n, range(n)

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            yield {"image": torch.full((1, 3, 2, 2), float(i))}


def _worker(rank, world, port, folder, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    from footprints_amd.parallel import DistContext
    from footprints_amd.training.train import TrainManager
    lines = []
    ctx = DistContext.from_env(use_cuda=False)
    assert ctx.world == 2 and ctx.rank == rank and ctx.active and ctx.is_main == (rank == 0)
    mm, step = _FakeModelManager(folder), _FakeStep()
    tm = TrainManager(_TaggedLoader(7), epochs=2, log=lines.append, log_freq=100, dist_context=ctx, model_manager=mm, train_step=step,
                      save_folder=folder)
    assert len(tm.train_loader) == 3                       # 7 global batches -> 3 per rank, the tail batch is dropped
    tm.train()
    q.put((rank, step.seen, lines, mm.saved, mm.scheduler.steps, tm.history["train"]))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_train_manager_shards_logs_and_saves_like_one_replica_of_n(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r = q.get(timeout=300)
        res[r[0]] = r
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # shards: rank r sees batches r, r + 2, r + 4 of every epoch; batch 6 (the tail) is nobody's
    assert res[0][1] == [0.0, 2.0, 4.0] * 2 and res[1][1] == [1.0, 3.0, 5.0] * 2
    # only rank 0 talks and saves; both stepped their scheduler once per epoch
    assert len(res[0][2]) >= 1 and res[0][2][0].startswith("Epoch 0 -- Batch 0 -- Loss") and res[1][2] == []
    assert res[0][3] == ["weights_0", "weights_1"] and res[1][3] == []
    assert sorted(os.listdir(tmp_path)) == ["weights_0", "weights_1"]
    assert res[0][4] == res[1][4] == 2
    # the logged average is the mean over the ranks (step 0: batches 0 and 1 -> 0.5), identical on both
    h0, h1 = res[0][5], res[1][5]
    assert h0 == h1 and h0[0][0] == 0 and abs(h0[0][1]["loss"] - 0.5) < 1e-12
    assert "0.5" in res[0][2][0]


def test_sharded_loader_and_single_process_context():
    from footprints_amd.parallel import DistContext, ShardedLoader
    base = _TaggedLoader(10)
    for world in (1, 2, 3, 4):
        seen = []
        for r in range(world):
            sh = ShardedLoader(base, r, world)
            got = [int(b["image"].flatten()[0]) for b in sh]
            assert len(got) == len(sh) == 10 // world
            seen += got
        assert sorted(seen) == list(range(10 // world * world))       # disjoint, covering, equal counts
    ctx = DistContext()
    assert not ctx.active and ctx.is_main and ctx.shard(base) is base and ctx.mean_losses({"loss": 2.0}) == {"loss": 2.0}
    ctx.barrier()
