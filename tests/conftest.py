import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle runs inside the tests: size torch's intra-op pool to the cores we really have
    import torch
    from oracle.cpu_threads import effective_cores
    torch.set_num_threads(min(effective_cores(), 32))


@pytest.fixture(scope="session")
def have_gpu():
    import torch
    return torch.cuda.is_available()


# ---- the parity ratio tables of a GPU session, assembled by the suite itself ---------------------------------------
_SESSION_T0 = [0.0]


def pytest_sessionstart(session):
    import time
    _SESSION_T0[0] = time.time()


def pytest_sessionfinish(session, exitstatus):
    """tests/test_gpu_parity_fullsize.py leaves one JSON per case under gpurun_out/parity/ (err(GPU) / err(CPU fp32) per parameter tensor,
    both against the float64 oracle); this writes the markdown table of the cases THIS session produced next to them --
    `parity_ratios.md` (both operand formats since round 5), copied into profiles/round<N>_parity_ratios.md by hand, never assembled by hand."""
    import glob
    import json
    out_dir = os.environ.get("FP_PARITY_DUMP", os.path.join(ROOT, "gpurun_out", "parity"))
    docs = []
    for path in sorted(glob.glob(os.path.join(out_dir, "*.json"))):
        try:
            if os.path.getmtime(path) >= _SESSION_T0[0] - 1.0:
                doc = json.load(open(path))
                if isinstance(doc, dict) and "case" in doc:          # (the directory also holds last_bn_decomposition.json)
                    docs.append(doc)
        except (OSError, ValueError):
            pass
    if not docs:
        return
    lines = ["# err(GPU) / err(CPU fp32) per parameter tensor, both against the float64 oracle -- written by the test session itself",
             "", "operand formats: exact = bf16x3 split, the default (footprints_amd/_format.py); fp16_pair = scaled fp16 pairs, opt-in.  Gates: tests/parity.py "
             "(median inside [0.4, 1.5]; natural-statistics case [0.2, 1.5]).  `forced` = the same tensors against the float64 oracle evaluated under the "
             "ENGINE's own ReLU decisions (tests/parity.py decision_forced_report): failures / max err / median err; evaluated for the headline case and "
             "wherever the single-run rule failed.", "",
             "| case | format | tensors | median | p90 | tensors > 2 | kink pixels removed | single-run failures | forced: failures / max / median | ReLU decisions != float64 | imposed decisions: flips / worst distance from the boundary in float64 (x channel RMS), engine; CPU fp32 run | worst tensors (ratio; err GPU / err CPU fp32) |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for d in docs:
        worst = ", ".join("%s %.1f (%.1e / %.1e)" % (t["tensor"].replace("encoder.", "enc.").replace("_decoder", "_dec"), t["ratio"], t["err_gpu"],
                                                     t["err_cpu32"]) for t in d.get("top6", [])[:3])
        f = d.get("decision_forced")
        forced = "%d / %.1e / %.1e" % (len(f["failures"]), f["max_err"], f["median_err"]) if f else ""
        fl = d.get("relu_decisions_differing_from_float64")
        im, rf = d.get("imposed_decisions"), d.get("cpu_fp32_decisions_vs_float64")
        imposed = ""
        if im:                                                   # round 6: the bound on what was imposed (tests/parity.py assert_decisions_at_roundoff)
            imposed = "%d / %.1e (pool %d / %.1e)" % (im["relu_flips"], im["relu_flip_worst_distance"], im["pool_flips"], im["pool_flip_worst_distance"])
            if rf:
                imposed += "; %d / %.1e" % (rf["relu_flips"], rf["relu_flip_worst_distance"])
        lines.append("| %s | %s | %s | %.2f | %.2f | %s | %s | %s | %s | %s | %s | %s |" % (
            d.get("case"), d.get("operand_format", ""), d.get("tensors"), d.get("median_ratio", float("nan")), d.get("p90_ratio", float("nan")),
            d.get("count_ratio_gt_2"), d.get("kink_pixels_removed", ""),
            len(d["single_run_rule_failures"]) if "single_run_rule_failures" in d else d.get("failures_under_single_run_rule", 0),
            forced, ("%d of %d" % (fl["engine"], fl["of"])) if fl else "", imposed, worst))
    try:
        with open(os.path.join(out_dir, "parity_ratios.md"), "w") as fh:
            fh.write("\n".join(lines) + "\n")
    except OSError:
        pass
