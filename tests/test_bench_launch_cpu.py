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


def test_compact_line_is_small_and_round_trips():
    """the driver parses the LAST stdout line and gave up on round 5's 34.6 KB one.  The line bench.py prints is
    compact_record(full record): built here from round 5's committed full record (the largest one this repo ever printed) plus a
    data-parallel config block, it must stay under 4 KB (hard cap 8 KB), round-trip through json and carry the contract keys."""
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, "profiles", "round5_bench_line_kitti_final_tree.json")) as fh:
        full = json.load(fh)
    assert len(json.dumps(full)) > 30000                           # the record that broke the driver
    full["config"]["gradient_exchange"] = {"transport": "rccl", "buckets": 7, "overlap_with_backward": True, "rccl_ranks": 8,
                                           "exposed_communication": {"exposed_ms": 0.41, "how": "x" * 500}, "allreduce_us": [{"bucket": "b" * 80}] * 7}
    full["step_bytes"] = {"algorithmic_gb": 15.2, "hbm_gb": 30.8, "traffic_ratio": 2.03, "kernel_launches": 582, "note": "n" * 900}
    full["roofline"]["mfma_random_operand_peak"] = {"tflops": 1850.0, "shader_mhz": 1814.0, "launches": 130, "frac_of_it": 0.46, "how": "h" * 300}
    line = bench.compact_record(full, "bench_detail.json")
    assert "\n" not in line and len(line) <= bench.COMPACT_LIMIT <= 4096 < 8192
    rec = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in rec, k
    assert rec["value"] == full["value"] and rec["ms_per_step"] == full["ms_per_step"] and rec["vs_baseline"] is None
    assert rec["config"]["workload"].startswith("KITTI 192x640 bs=12") and "model" not in rec["config"]
    rf = rec["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_kernel_us", "launches_per_step"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = rec["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "single_thread" in cb and len(cb["sample"]) <= 160
    assert rec["config"]["gradient_exchange"]["rccl_ranks"] == 8 and rec["step"]["traffic_ratio"] == 2.03
    assert rf["mfma_random_operand_peak"] == {"tflops": 1850.0, "shader_mhz": 1814.0, "frac_of_it": 0.46}      # the measured ceiling on data, beside the nominal peak
    # a record with nothing optional in it (kernel events and CPU baseline switched off) still makes a valid line
    bare = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data", "config")}
    assert json.loads(bench.compact_record(bare))["value"] == full["value"]


def test_default_run_skips_the_lab_notebook_legs():
    """the default command line (what the driver runs) has the sustained leg and the second-format child switched off"""
    sys.path.insert(0, ROOT)
    import bench
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'ap.add_argument("--sustain", type=float, default=0.0' in src
    assert 'ap.add_argument("--other-format", dest="other_format", action="store_true"' in src
    assert "print(compact_record(out, write_detail(out)), flush=True)" in src and "print(json.dumps(out)" not in src


def test_whole_step_algorithmic_bytes_reproduce_the_survey_figures():
    """VERDICT r5 "Next" 7: the whole-step fused-minimum byte count behind `step.traffic_ratio`.  The decoder and loss families must land on
    SURVEY.md section 8(d) / BASELINE.md section 3 (3.544 / 7.018 / 0.224 GB at 12x192x640, 3.155 / 6.247 / 0.199 at 4x512x640; the survey's
    totals sit 2 % above the sum of its own per-convolution rows, which this function reproduces to the megabyte: 1 736.5 MB per decoder
    forward), the encoder follows the same rule, backward = 2 x forward by construction, and Adam moves 7 floats per parameter."""
    sys.path.insert(0, ROOT)
    import bench
    k = bench.step_algorithmic_bytes(12, 192, 640)
    m = bench.step_algorithmic_bytes(4, 512, 640)
    assert abs(k["decoders_fwd"] - 2 * 1.7365) < 0.002                     # the survey table's own rows, summed
    for got, want in ((k["decoders_fwd"], 3.544), (k["decoders_bwd"], 7.018), (k["loss"], 0.224), (m["decoders_fwd"], 3.155),
                      (m["decoders_bwd"], 6.247), (m["loss"], 0.199)):
        assert abs(got - want) <= 0.03 * want, (got, want)
    for d in (k, m):
        assert abs(d["encoder_bwd"] - 2 * d["encoder_fwd"]) < 0.002 and abs(d["decoders_bwd"] - 2 * d["decoders_fwd"]) < 0.002
        assert abs(d["adam"] - 31012944 * 28 / 1e9) < 0.001
        assert abs(d["total"] - sum(v for kk, v in d.items() if kk != "total")) < 0.005
    assert 0.8 < k["encoder_fwd"] < 1.1 and 13.5 < k["total"] < 15.0
