#!/bin/bash
# round 6, batch l: convolutions storing through the pixel shuffle -- tests, model parity, then same-box A/B of the step
set -o pipefail
mkdir -p gpurun_out/r06l
timeout -k 10 400 python -m pytest tests/test_gpu_conv_shuffle.py -x -q -m gpu > gpurun_out/r06l/tests.log 2>&1 || { tail -40 gpurun_out/r06l/tests.log; exit 1; }
tail -3 gpurun_out/r06l/tests.log
timeout -k 10 500 python -m pytest tests/test_gpu_model.py tests/test_gpu_c16.py -x -q -m gpu > gpurun_out/r06l/tests2.log 2>&1 || { tail -40 gpurun_out/r06l/tests2.log; exit 1; }
tail -3 gpurun_out/r06l/tests2.log
for i in 1 2; do
  timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-inference --detail gpurun_out/r06l/new$i.json > gpurun_out/r06l/new$i.line 2> gpurun_out/r06l/new$i.err || exit 1
  EBFI_DEV=1 EBFI_NO_CONV_SHUFFLE=1 timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-inference --detail gpurun_out/r06l/old$i.json > gpurun_out/r06l/old$i.line 2> gpurun_out/r06l/old$i.err || exit 1
done
python - <<'PY'
import json
for t in ("new1","old1","new2","old2"):
    d=json.loads(open("gpurun_out/r06l/%s.line"%t).read().strip().splitlines()[-1])
    print(t, d["ms_per_step"], d["value"])
PY
