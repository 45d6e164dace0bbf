#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4u; mkdir -p $O
cd $R
T="tests/test_gpu_network.py::test_g2_decoder_golden_through_engine"
for cfg in "" "FP_PACK_LAZY32=0" "FP_PACK_TILED=0" "FP_HP_IGEMM=0" "FP_HP=0" "FP_PACK_LAZY32=0 FP_PACK_TILED=0"; do
  echo "== cfg: $cfg" >> $O/bisect.txt
  env $cfg timeout 300 python -m pytest "$T" -x -q 2>&1 | grep -E "passed|failed|AssertionError" | head -3 >> $O/bisect.txt
done
echo "== whole file default" >> $O/bisect.txt
timeout 600 python -m pytest tests/test_gpu_network.py -q 2>&1 | grep -E "passed|failed|AssertionError" | head -5 >> $O/bisect.txt
cat $O/bisect.txt
