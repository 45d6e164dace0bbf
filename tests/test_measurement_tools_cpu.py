"""CPU: the measurement tooling around bench.py on synthetic inputs -- the sysfs sampler degrades gracefully on a box without an
amdgpu device node, and the step-level counter summary (scripts/pmc_step.py) turns rocprofv3 counter CSVs into per-step figures with the documented
unit corrections (FETCH_SIZE x 2 on gfx950, MFMA busy = (SQ_VALU_MFMA_BUSY_CYCLES / 1024) / (GRBM_GUI_ACTIVE / 8))."""
import csv
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpu_sampler_without_device_nodes_reports_nothing_instead_of_failing():
    sys.path.insert(0, ROOT)
    import bench
    s = bench.GpuSampler(index=63, period=0.01).start()          # no such card on any box
    time.sleep(0.05)
    s.stop()
    out = s.summary()
    assert out["sclk_mhz"] is None and out["power_w"] is None and out["busy_percent"] is None and "no amdgpu device node" in out["source"]


def _write_csv(d, rows):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "pmc_counter_collection.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=["Kernel_Name", "Counter_Name", "Counter_Value"])
        w.writeheader()
        for r in rows:
            w.writerow(dict(zip(w.fieldnames, r)))


def test_pmc_step_summary_applies_the_documented_corrections(tmp_path):
    f, wdir, q = str(tmp_path / "f"), str(tmp_path / "w"), str(tmp_path / "q")
    k1, k2 = "void conv3x3_tile_bf3_kernel<8, 16, 64, 2, 2, false, false, 3, false, false>(Tile3Args)", "adam_kernel(float*)"
    steps, warm = 2, 1                                           # three steps in the CSV
    _write_csv(f, [(k1, "FETCH_SIZE", 1000.0)] * 3 + [(k2, "FETCH_SIZE", 500.0)] * 3 + [("__amd_rocclr_copyBuffer", "FETCH_SIZE", 9e9)])
    _write_csv(wdir, [(k1, "WRITE_SIZE", 400.0)] * 3 + [(k2, "WRITE_SIZE", 300.0)] * 3)
    _write_csv(q, [(k1, "SQ_VALU_MFMA_BUSY_CYCLES", 1024.0 * 500)] * 3 + [(k1, "GRBM_GUI_ACTIVE", 8.0 * 1000)] * 3 + [(k2, "GRBM_GUI_ACTIVE", 8.0 * 1000)] * 3 +
               [(k1, "SQ_INSTS_MFMA", 16000.0)] * 3)
    out = str(tmp_path / "step.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_step.py"), "kitti", "exact", f, wdir, q, str(steps), str(warm), out, "10.0"],
                       capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-800:]
    d = json.load(open(out))
    assert d["operand_format"] == "exact" and d["steps_in_the_csv"] == 3 and len(d["kernel_source_sha16"]) == 16
    assert d["hbm_bytes_per_step"] == {"fetch": 1500 * 1024 * 2, "write": 700 * 1024, "total": 1500 * 1024 * 2 + 700 * 1024}     # the runtime's own copies are not the library's
    assert d["mfma"]["busy_fraction_of_serial_kernel_time"] == 0.25 and d["mfma"]["top_kernels"][0]["busy_fraction_of_its_own_time"] == 0.5
    assert d["kernel_launches_per_step"] == 2.0
    gbs = d["hbm_gb_per_s_over_the_concurrent_step"]
    assert abs(gbs["gb_per_s"] - round((1500 * 1024 * 2 + 700 * 1024) / 10.0 / 1e6, 1)) < 1e-9
