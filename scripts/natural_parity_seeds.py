"""GPU box: the natural-statistics / wide-range parity case (tests/test_gpu_parity_fullsize.py) over several image seeds and BOTH operand
formats, under the SINGLE-RUN rule (err(GPU) <= 4 x max(err of ONE fp32 CPU run, stage median)): does the fp16-pair path lose the "ReLU
lottery" of that chaotic case more often than the exact split?  

    python scripts/natural_parity_seeds.py [n_seeds=8] [B=4]          -> gpurun_out/parity/natural_seeds.md

Per seed the float64 and float32 CPU oracles run once (parent process) and both formats are measured against them in child processes
(the operand format is fixed when footprints_amd.engine is imported: FP_HP)."""
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
H, W = 192, 640


def child(path):
    from tests.parity import anchored_report
    from tests.test_gpu_parity_fullsize import _gpu_step, _wide_range_state
    doc = torch.load(path)
    P, B = _wide_range_state()
    _, _, _, g_gpu = _gpu_step(P, B, doc["batch"])
    bad, rows = anchored_report(g_gpu, doc["g32"], doc["g64"])
    ratios = sorted(r for r, *_ in rows)
    print(json.dumps({"median": ratios[len(ratios) // 2], "gt2": sum(r > 2 for r in ratios), "gt4": sum(r > 4 for r in ratios),
                      "worst": ratios[-1], "worst_tensor": rows[0][1], "fail_single_run": len(bad)}))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    Bn = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    from oracle.cpu_threads import effective_cores
    torch.set_num_threads(min(effective_cores(), 32))
    from tests.parity import TIE_SIGMA, oracle_grads, tie_free_batch
    from tests.test_gpu_parity_fullsize import _natural_batch, _wide_range_state
    P, B = _wide_range_state()
    rows = []
    for seed in range(n):
        t0 = time.time()
        batch = _natural_batch(Bn, H, W, seed=31 + 97 * seed)
        _, _, g64, _, used = oracle_grads(P, B, batch, torch.float64, fix_batch=lambda b, o: tie_free_batch(b, o, tie_sigma=TIE_SIGMA)[0])
        _, _, g32, _, _ = oracle_grads(P, B, used, torch.float32)
        path = "/tmp/natural_seed.pt"
        torch.save({"batch": used, "g64": g64, "g32": g32}, path)
        t_cpu = time.time() - t0
        for fmt, hp in (("fp16 pairs", "1"), ("exact split", "0")):
            env = dict(os.environ, FP_HP=hp)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", path], env=env, capture_output=True, text=True, timeout=900)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            r = json.loads(line[-1]) if line else {"error": (p.stderr or "")[-200:]}
            r.update(seed=seed, format=fmt)
            rows.append(r)
            print(seed, fmt, r, "cpu %.0f s" % t_cpu, flush=True)
    out = os.path.join(ROOT, "gpurun_out", "parity")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "natural_seeds.md"), "w") as fh:
        fh.write("# natural-statistics / wide-range case (%d x %d x %d), single-run rule, per image seed and operand format\n\n" % (Bn, H, W))
        fh.write("| seed | format | median ratio | tensors > 2 | tensors > 4 | worst ratio | worst tensor | failures under the single-run rule |\n|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            if "error" in r:
                fh.write("| %d | %s | error: %s |\n" % (r["seed"], r["format"], r["error"].replace("\n", " ")[:120]))
            else:
                fh.write("| %d | %s | %.2f | %d | %d | %.1f | %s | %d |\n" % (r["seed"], r["format"], r["median"], r["gt2"], r["gt4"], r["worst"], r["worst_tensor"], r["fail_single_run"]))
        for fmt in ("fp16 pairs", "exact split"):
            ok = [r for r in rows if r["format"] == fmt and "error" not in r]
            if ok:
                fh.write("\n%s: seeds with failures under the single-run rule %d of %d; mean tensors > 2: %.1f; worst ratio over all seeds %.1f\n" % (
                    fmt, sum(r["fail_single_run"] > 0 for r in ok), len(ok), sum(r["gt2"] for r in ok) / len(ok), max(r["worst"] for r in ok)))
    print(open(os.path.join(out, "natural_seeds.md")).read())


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        main()
