#!/bin/bash
# GPU box: which hardware queue does each stream's kernels land on?  AMD_LOG_LEVEL=4 prints one line per dispatch with the software
# and hardware queue; two eager steps per mode, the interesting lines (bounded) kept under gpurun_out/.
out=${1:-gpurun_out/qmap}; mkdir -p $out
for mode in single comm_after_engine; do
  AMD_LOG_LEVEL=4 FP_PLAN=0 PROBE_STEPS=2 timeout 300 python scripts/dp_tax_probe2.py $mode > $out/$mode.stdout 2> $out/$mode.raw
  wc -l < $out/$mode.raw > $out/$mode.nlines
  grep -i -E "hwq|hardware queue|acquire|queue" $out/$mode.raw | tail -c 1500000 > $out/$mode.queue_lines
  head -c 100000 $out/$mode.raw > $out/$mode.head
  tail -c 300000 $out/$mode.raw > $out/$mode.tail
  rm -f $out/$mode.raw
done
