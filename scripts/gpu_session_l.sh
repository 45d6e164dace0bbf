#!/bin/bash
set +e
export TMPDIR=/tmp
O=gpurun_out/r2l
mkdir -p $O
( timeout 500 python bench.py --dump-kernels $O/kernels_kitti.json ) > $O/bench_kitti.json 2> $O/bench_kitti.err; echo "kitti rc=$?"; tail -2 $O/bench_kitti.err
( timeout 500 python bench.py --workload matterport --dump-kernels $O/kernels_mp.json ) > $O/bench_mp.json 2> $O/bench_mp.err; echo "mp rc=$?"; tail -2 $O/bench_mp.err
( timeout 300 python bench.py --force-dist --steps 20 --warmup 5 --no-cpu-baseline ) > $O/bench_forcedist.json 2> $O/bench_forcedist.err; echo "forcedist rc=$?"; tail -2 $O/bench_forcedist.err
python - <<'PY'
import json
for f in ("kitti","mp","forcedist"):
    try:
        d=json.load(open("gpurun_out/r2l/bench_%s.json"%f))
    except Exception as e:
        print(f,"ERR",e); continue
    r=d.get("roofline") or {}
    print(f, d["value"], d["ms_per_step"], d["fwd_ms_per_img"], d["decoder_backward"]["ms"], r.get("kernel","")[:30], r.get("frac"), (r.get("traffic") or {}).get("ratio_to_algorithmic"), d["config"]["parallelism"])
PY
python scripts/hbm_microbench.py > $O/hbm_kernels.txt 2>&1; tail -3 $O/hbm_kernels.txt
