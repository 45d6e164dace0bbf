#!/bin/bash
set +e
export TMPDIR=/tmp
O=gpurun_out/r2h
mkdir -p $O
( timeout 1800 python -m pytest tests -q -m gpu --durations=10 ) > $O/pytest_all.log 2>&1
echo "pytest rc=$?"; tail -18 $O/pytest_all.log
( timeout 400 python bench.py --dump-kernels $O/kernels_kitti.json ) > $O/bench_kitti.json 2> $O/bench_kitti.err
echo "bench rc=$?"; tail -2 $O/bench_kitti.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r2h/bench_kitti.json"))
print(d["value"], d["ms_per_step"], d["fwd_ms_per_img"], d["decoder_backward"]["ms"])
print(d["device_data_path"])
print(d["cpu_baseline"])
r=d["roofline"]; print(r["kernel"], r["frac"], r["exclusive_ms_per_step"], r["traffic"])
PY
