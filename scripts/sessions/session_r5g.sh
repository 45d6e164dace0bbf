#!/bin/bash
# GPU box, round 5 session G: the whole GPU suite on the final tree, smoke, and the driver's bench command.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5g; mkdir -p $O
cd $R
t0=$(date +%s)
timeout 1800 python -m pytest tests -m gpu -q --durations=15 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? t=$(( $(date +%s)-t0 ))" > $O/summary.txt
tail -30 $O/pytest_gpu.log
cp -r gpurun_out/parity $O/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$? t=$(( $(date +%s)-t0 ))" >> $O/summary.txt
tail -2 $O/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench.err; echo "bench rc=$? t=$(( $(date +%s)-t0 ))" >> $O/summary.txt
python -c "
import json; d=json.loads(open('$O/bench_driver_cmd.json').read().strip().splitlines()[-1])
print('value', d['value'], d['ms_per_step'], d['dtype'][:50]); print('sustained', d['sustained']['img_per_s'], d['sustained']['shader_clock_mhz'])
print('fp16_pair', d['fp16_pair'].get('value'), d['fp16_pair'].get('error')); print('roofline', d['roofline']['frac'], d['roofline']['traffic_ratio'], d['roofline']['step_counters'].get('mfma_busy_fraction_of_serial_kernel_time'))
print('cpu', d['cpu_baseline']['value'])
"
cat $O/summary.txt
