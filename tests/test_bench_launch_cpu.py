"""bench.py's multi-rank launch path, without a GPU (VERDICT r2 "Next" 1a): `python bench.py --gpus N` started plainly must become
the launcher of its N ranks, the ranks must rendezvous on 127.0.0.1, and rank 0 must print exactly ONE JSON line with n_gpus = N;
the driver's own form (`python -m torch.distributed.run ... bench.py --gpus N`) must do the same.  --dry-run-dist stops after the
rendezvous + one collective (the product has no CPU compute path; the GPU legs run in tests/test_gpu_dp.py)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_lines(stdout):
    return [json.loads(l) for l in stdout.splitlines() if l.startswith("{")]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_plain_invocation_spawns_its_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-dist"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["ranks_sum"] == 3.0      # both ranks took part in the collective


def test_driver_form_torch_distributed_run():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-dist"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2


def test_single_rank_needs_a_gpu_and_says_so():
    import torch
    if torch.cuda.is_available():
        return
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "needs a GPU" in (p.stderr + p.stdout)
