"""Step-level counters (VERDICT r4 "Next" 5: north_star asks for "rocprof HBM GB/s and MFMA utilisation" of the STEP, not of a microbenchmark).

Three counters-only rocprofv3 passes over scripts/step_loop.py (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; the SQ pass carries
the MFMA counters), every dispatch of the measured steps summed:

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o pmc -- python scripts/step_loop.py kitti 2 1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o pmc -- python scripts/step_loop.py kitti 2 1
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d /tmp/pq -o pmc -- python scripts/step_loop.py kitti 2 1
    python scripts/pmc_step.py kitti <operand format> /tmp/pf /tmp/pw /tmp/pq <steps counted> <warm-up steps> out.json [step_ms from bench.py]

Units (MI355X_MICROARCH.md): FETCH_SIZE / WRITE_SIZE in KB, FETCH_SIZE x2 on gfx950 (it reports half of wide coalesced reads), WRITE_SIZE as
reported; SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE in cycles summed over the 8 XCDs.  Under
counter collection rocprofv3 serialises the dispatches, so GRBM_GUI_ACTIVE / 8 is the serial kernel time of the step in shader cycles and
    MFMA busy % = (sum SQ_VALU_MFMA_BUSY_CYCLES / 1024) / (sum GRBM_GUI_ACTIVE / 8)
is the share of the step's kernel time the matrix pipes work, averaged over the SIMDs -- the step-level counterpart of the per-kernel figure in
profiles/round4_pmc_sq_hp.txt.  HBM GB/s = bytes per step / step time (the concurrent step of bench.py when given, and the serial kernel time
derived from GRBM_GUI_ACTIVE at the sustained clock otherwise)."""
import csv
import glob
import hashlib
import json
import os
import sys


def collect(d):
    per = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                a = per.setdefault(row["Counter_Name"], {})
                k = row["Kernel_Name"].replace("(anonymous namespace)::", "")
                e = a.setdefault(k, [0, 0.0])
                e[0] += 1
                e[1] += float(row["Counter_Value"])
    return per


def main():
    wl, fmt, df, dw, dq, steps, warm, out = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], int(sys.argv[6]), int(sys.argv[7]), sys.argv[8]
    step_ms = float(sys.argv[9]) if len(sys.argv) > 9 else None
    total = steps + warm                       # every step of the process is in the csv (warm-up steps run the same kernels): per-step = sum / total
    f, w, q = collect(df).get("FETCH_SIZE", {}), collect(dw).get("WRITE_SIZE", {}), collect(dq)
    lib_only = lambda d: {k: v for k, v in d.items() if not k.startswith("__amd_rocclr") and "at::native" not in k}
    fetch = sum(v[1] for v in lib_only(f).values()) * 1024 * 2 / total
    write = sum(v[1] for v in lib_only(w).values()) * 1024 / total
    mfma_busy = sum(v[1] for v in lib_only(q.get("SQ_VALU_MFMA_BUSY_CYCLES", {})).values()) / total
    gui = sum(v[1] for v in lib_only(q.get("GRBM_GUI_ACTIVE", {})).values()) / total
    insts = sum(v[1] for v in lib_only(q.get("SQ_INSTS_MFMA", {})).values()) / total
    launches = sum(v[0] for v in lib_only(q.get("GRBM_GUI_ACTIVE", {})).values()) / total
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hsh = hashlib.sha256()
    for name in ("conv3x3_tile_bf3.hip", "fp_common.h"):
        with open(os.path.join(root, "footprints_amd", "csrc", name), "rb") as fh:
            hsh.update(fh.read())
    busy = (mfma_busy / 1024.0) / (gui / 8.0) if gui > 0 else None
    top = sorted(((v[1] / total, k) for k, v in lib_only(q.get("SQ_VALU_MFMA_BUSY_CYCLES", {})).items()), reverse=True)[:8]
    gq = lib_only(q.get("GRBM_GUI_ACTIVE", {}))
    res = {"workload": wl, "operand_format": fmt, "kernel_source_sha16": hsh.hexdigest()[:16], "steps_in_the_csv": total,
           "kernel_launches_per_step": round(launches, 1),
           "hbm_bytes_per_step": {"fetch": round(fetch), "write": round(write), "total": round(fetch + write)},
           "mfma": {"busy_cycles_per_step_all_simds": round(mfma_busy), "gui_active_cycles_per_step_all_xcds": round(gui), "mfma_instructions_per_step": round(insts),
                    "busy_fraction_of_serial_kernel_time": round(busy, 4) if busy is not None else None,
                    "top_kernels": [{"kernel": k[:120], "busy_cycles_per_step": round(c),
                                     "busy_fraction_of_its_own_time": round((c / 1024.0) / (gq[k][1] / total / 8.0), 4) if k in gq and gq[k][1] > 0 else None}
                                    for c, k in top]},
           "_units": "per training step (sum over every dispatch of the library's kernels in `python scripts/step_loop.py %s %d %d` / %d steps); fetch = FETCH_SIZE KB x "
                     "1024 x 2 (gfx950), write = WRITE_SIZE KB x 1024; MFMA busy = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs), dispatches "
                     "serialised by the counter collection" % (wl, steps, warm, total)}
    if step_ms:
        res["hbm_gb_per_s_over_the_concurrent_step"] = {"step_ms": step_ms, "gb_per_s": round((fetch + write) / step_ms / 1e6, 1),
                                                         "fraction_of_8_tb_per_s": round((fetch + write) / step_ms / 1e6 / 8000.0, 4)}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
