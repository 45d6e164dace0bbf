"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- they do not fit one pass on gfx950) over scripts/step_loop.py
into profiles/round<N>_pmc_hbm_<workload>.json, the file bench.py's `roofline.traffic` is filled from.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o pmc -- python scripts/step_loop.py kitti 2 1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o pmc -- python scripts/step_loop.py kitti 2 1
    python scripts/pmc_hbm.py kitti gpurun_out/pmc_f gpurun_out/pmc_w [kernels.json from bench.py --dump-kernels] profiles/round2_pmc_hbm_kitti.json

Units / corrections (MI355X_MICROARCH.md, HBM section): the counters are in KB; on gfx950 FETCH_SIZE reports exactly 1/2 of the
bytes of wide coalesced reads -> x2; WRITE_SIZE is uncalibrated and is given as reported.  Per kernel family: average per launch."""
import csv
import glob
import json
import os
import sys

KERNEL_TO_ENTRY = [("wgrad3x3_hp_pf_kernel", "conv_wgrad_hp"), ("wgrad3x3_bf3_v", "conv_wgrad_bf3"), ("wgrad_up2_phase_bf3_kernel", "conv_up2_phase_wgrad_bf3"),
                   ("conv3x3_tile_bf3_kernel", "conv3x3_bf3"), ("up2_phase_fwd_bf3_kernel", "conv_up2_phase_fwd_bf3"),
                   ("up2_phase_dgrad_bf3_kernel", "conv_up2_phase_dgrad_bf3"), ("igemm_hp_kernel", "conv_igemm_hp"), ("igemm_kernel", "conv_igemm"), ("stem_tile_kernel", "conv_igemm"),
                   ("wgrad3x3_tile_kernel", "conv_wgrad"), ("wgrad_kernel", "conv_wgrad")]
AUX = ("wgrad_reduce_bias_t_kernel", "wgrad_reduce_bias_kernel", "wgrad_reduce_kernel", "wgrad_bias_reduce_kernel", "splitk_reduce_kernel", "up2_wgrad_sum_kernel", "up2_wgrad_uncollapse_kernel",
       "up2_wgrad_bias_reduce_kernel")


def collect(d, counter):
    per = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                name = row["Kernel_Name"]
                key = None
                for frag in AUX:
                    if frag in name:
                        key = "aux:" + frag
                        break
                if key is None:
                    for frag, entry in KERNEL_TO_ENTRY:
                        if frag in name:
                            key = entry
                            break
                    # fp16-pair instantiations of the same kernel templates (NP = 2 / HP = true) are their own entry points
                    hp = ((key == "conv3x3_bf3" and (", 2, true>" in name or ", 2, true, " in name)) or
                          (key == "conv_wgrad_bf3" and ", 2>(" in name) or (key == "conv_up2_phase_wgrad_bf3" and "<2>(" in name) or
                          (key in ("conv_up2_phase_fwd_bf3", "conv_up2_phase_dgrad_bf3") and ", 2>(" in name))
                    if hp:
                        key = {"conv3x3_bf3": "conv3x3_hp", "conv_wgrad_bf3": "conv_wgrad_hp", "conv_up2_phase_wgrad_bf3": "conv_up2_phase_wgrad_hp",
                               "conv_up2_phase_fwd_bf3": "conv_up2_phase_fwd_hp", "conv_up2_phase_dgrad_bf3": "conv_up2_phase_dgrad_hp"}[key]
                # round 5: the ring weight gradient and the flattened split-operand GEMM also run with exact bf16x3 operands (last template argument 3)
                if key == "conv_wgrad_hp" and name.rstrip().split("(")[0].endswith(", 3>"):
                    key = "conv_wgrad_bf3"
                if key == "conv_igemm_hp" and name.rstrip().split("(")[0].endswith(", 3>"):
                    key = "conv_igemm_bf3"
                if key is None:
                    continue
                a = per.setdefault(key, [0, 0.0])
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return per


def main():
    wl, df, dw = sys.argv[1], sys.argv[2], sys.argv[3]
    kj = sys.argv[4] if len(sys.argv) > 5 else None
    out = sys.argv[-1]
    alg = {}
    if kj:
        for g in json.load(open(kj))["groups"]:
            alg[g["entry_point"]] = g["algorithmic_mb_per_launch"] * 1e6
    f, w = collect(df, "FETCH_SIZE"), collect(dw, "WRITE_SIZE")
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hsh = hashlib.sha256()
    for name in ("conv3x3_tile_bf3.hip", "fp_common.h"):        # the same digest bench.py computes (kernel_source_digest): ties this file to its build
        with open(os.path.join(root, "footprints_amd", "csrc", name), "rb") as fh:
            hsh.update(fh.read())
    res = {"kernel_source_sha16": hsh.hexdigest()[:16],
           "_units": "bytes per launch, averaged over every launch of the kernel family in `python scripts/step_loop.py %s` (train steps only); "
                     "fetch = FETCH_SIZE KB x 1024 x 2 (gfx950 reports half of wide coalesced reads), write = WRITE_SIZE KB x 1024 (uncalibrated)" % wl}
    for key in sorted(set(f) | set(w)):
        nf, vf = f.get(key, (0, 0.0))
        nw, vw = w.get(key, (0, 0.0))
        e = {"launches_counted": max(nf, nw), "fetch_bytes_per_launch": round(vf / max(nf, 1) * 1024 * 2), "write_bytes_per_launch": round(vw / max(nw, 1) * 1024)}
        if key in alg:
            e["algorithmic_bytes_per_launch"] = round(alg[key])
            e["ratio_to_algorithmic"] = round((e["fetch_bytes_per_launch"] + e["write_bytes_per_launch"]) / max(alg[key], 1.0), 3)
        res[key] = e
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
