#!/bin/bash
# round 6, batch w: inference -- the fused KernelConv -> FAC kernel takes the concatenation's parts; tests, then configs 2 / 5 new vs old
set -o pipefail
mkdir -p gpurun_out/r06w
timeout -k 10 900 python -m pytest tests/test_gpu_fac.py tests/test_gpu_entrypoints.py -x -q -m gpu > gpurun_out/r06w/tests.log 2>&1 || { tail -40 gpurun_out/r06w/tests.log; exit 1; }
tail -2 gpurun_out/r06w/tests.log
for i in 1 2; do
  for cfg in "--batch 4 --height 256 --width 256" "--batch 8 --height 720 --width 1280 --num_ts 8"; do
    echo -n "new $cfg : "; timeout -k 10 300 python ebfi-be_amd/infer_ours.py --rand-init $cfg 2>/dev/null | tail -1 | cut -c1-100
    echo -n "old $cfg : "; EBFI_DEV=1 EBFI_NO_CAT16=1 timeout -k 10 300 python ebfi-be_amd/infer_ours.py --rand-init $cfg 2>/dev/null | tail -1 | cut -c1-100
  done
done
