"""GPU box: take the data-parallel branch's fixed cost apart (profiles/round3_notes.md).

    python scripts/dp_tax_bisect.py            runs every variant below in a fresh process and prints one table
    python scripts/dp_tax_bisect.py <variant>  one variant (what the parent spawns)

Variants (KITTI 12x192x640 training step, 8 warm-up + 30 timed steps, ms/step from HIP events):
  single          TrainStep(distributed=False), no process group
  streams2        single + two extra streams that each run one tiny kernel per step (hardware-queue sharing without any RCCL)
  pg_nccl         torch "nccl" process group created (device_id given) + one all-reduce, then the single-GPU step
  torch           forced world-of-one through torch.distributed collectives on a "nccl" group (round 2's path)
  rccl_own        forced world-of-one through fp_comm_* (no process group at all), all-reduces on a dedicated stream
  rccl_dwg0       ... on the mask decoder's weight-gradient stream (idle once the decoders are done)
  rccl_noplan     rccl_own with FP_PLAN=0 (every step issued from Python)
  rccl_nooverlap  rccl_own with FP_DP_OVERLAP=0 (all buckets after the backward pass)
  rccl_nocoll     rccl_own with the collectives themselves skipped (events and waits only)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = {
    "single": {},
    "streams2": {},
    "pg_nccl": {},
    "torch": {"FP_DP_FORCE": "1", "FP_DP_TRANSPORT": "torch"},
    "rccl_own": {"FP_DP_FORCE": "1", "FP_DP_TRANSPORT": "rccl", "FP_DP_COMM_STREAM": "own"},
    "rccl_dwg0": {"FP_DP_FORCE": "1", "FP_DP_TRANSPORT": "rccl", "FP_DP_COMM_STREAM": "dwg0"},
    "rccl_noplan": {"FP_DP_FORCE": "1", "FP_DP_TRANSPORT": "rccl", "FP_PLAN": "0"},
    "rccl_nooverlap": {"FP_DP_FORCE": "1", "FP_DP_TRANSPORT": "rccl", "FP_DP_OVERLAP": "0"},
    "rccl_nocoll": {"FP_DP_FORCE": "1", "FP_DP_TRANSPORT": "rccl"},
}


def run(variant):
    import torch
    torch.cuda.set_device(0)
    dist_on = variant in ("torch",) or variant.startswith("rccl")
    if variant in ("pg_nccl", "torch"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        dist.all_reduce(torch.ones(1024, device="cuda"))
        torch.cuda.synchronize()
    from footprints_amd import ops, parallel
    from footprints_amd.model_manager import ModelManager
    from footprints_amd.training.train import SEED, TrainStep, synthetic_batch
    if variant == "rccl_nocoll":
        parallel.Communicator.allreduce = lambda self, t, stream: None
    torch.manual_seed(SEED)
    mm = ModelManager(use_cuda=True)
    ts = TrainStep(mm.model, mm.optimiser, distributed=dist_on)
    batch = synthetic_batch(12, 192, 640, "cuda")
    extra = [torch.cuda.Stream(), torch.cuda.Stream()] if variant == "streams2" else []
    tiny = torch.zeros(64, device="cuda")

    def step():
        ts(batch)
        for s in extra:
            with ops.on_stream(s):
                ops.fill(tiny, 1.0)
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        step()
    e1.record()
    torch.cuda.synchronize()
    tr = ts.reducer.transport if ts.reducer is not None else "-"
    print("RESULT %s %.3f transport=%s loss=%.5f" % (variant, e0.elapsed_time(e1) / 30, tr, float(ts.losses[20])), flush=True)


def main():
    if len(sys.argv) > 1:
        return run(sys.argv[1])
    rows = []
    for v, env in VARIANTS.items():
        e = dict(os.environ)
        e.update(env)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), v], env=e, capture_output=True, text=True, timeout=300)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
        rows.append(line[0] if line else "RESULT %s FAILED rc=%d %s" % (v, p.returncode, (p.stderr or "")[-400:].replace("\n", " | ")))
        print(rows[-1], flush=True)


if __name__ == "__main__":
    main()
