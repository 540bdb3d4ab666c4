#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r06s
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "thin_layer" > gpurun_out/r06s/tests.log 2>&1 || { tail -40 gpurun_out/r06s/tests.log; exit 1; }
tail -2 gpurun_out/r06s/tests.log
timeout -k 10 200 python tools/shiftdiag.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06s/shiftdiag.txt
