#!/bin/bash
# round 6, batch v: the image of cat([ev, FrameTensor]) from its two parts -- tests, model parity, same-box A/B
set -o pipefail
mkdir -p gpurun_out/r06v
timeout -k 10 900 python -m pytest tests/test_gpu_fac.py tests/test_gpu_c16.py tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r06v/tests.log 2>&1 || { tail -40 gpurun_out/r06v/tests.log; exit 1; }
tail -2 gpurun_out/r06v/tests.log
for i in 1 2; do
  timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-inference --no-cpu-baseline --detail gpurun_out/r06v/new$i.json > gpurun_out/r06v/new$i.line 2> gpurun_out/r06v/new$i.err || exit 1
  EBFI_DEV=1 EBFI_NO_CAT16=1 timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-inference --no-cpu-baseline --detail gpurun_out/r06v/old$i.json > gpurun_out/r06v/old$i.line 2> gpurun_out/r06v/old$i.err || exit 1
done
python - <<'PY'
import json
for t in ("new1","old1","new2","old2"):
    d=json.loads(open("gpurun_out/r06v/%s.line"%t).read().strip().splitlines()[-1])
    print(t, d["ms_per_step"], d["value"])
PY
