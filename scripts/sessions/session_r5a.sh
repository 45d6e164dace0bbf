#!/bin/bash
# GPU box, round 5 session A: the whole GPU suite on the new default (exact bf16x3 operands; both formats in the full-size parity cases),
# then the default bench line (exact = value; fp16_pair leg; sustained legs).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5a; mkdir -p $O
cd $R
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --durations=25 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? t=$(( $(date +%s)-t0 ))" > $O/summary.txt
tail -40 $O/pytest_gpu.log
cp -r gpurun_out/parity $O/ 2>/dev/null
timeout 600 python bench.py > $O/bench_kitti.json 2> $O/bench_kitti.err; echo "bench rc=$? t=$(( $(date +%s)-t0 ))" >> $O/summary.txt
cat $O/summary.txt
python - <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r5a/bench_kitti.json").read().strip().splitlines()[-1])
    print("value", d["value"], d["ms_per_step"], d["dtype"][:40]); print("sustained", d.get("sustained"))
    print("fp16_pair", {k: d.get("fp16_pair",{}).get(k) for k in ("value","ms_per_step","error")})
    for g in d["roofline"]["groups"]: print(g["entry_point"], g["launches_per_step"], g["exclusive_ms_per_step"], g.get("kernel_only_ms_per_step"), g.get("frac_kernel_only"))
    for r in d["kernels"]["serial"][:24]: print(r)
except Exception as e: print("parse failed", e)
P
