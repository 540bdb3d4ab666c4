#!/bin/bash
# round 6, batch j: the squeeze-excite gate through the pixel shuffle -- tests, then same-box A/B of the step (switch off = the old path)
set -o pipefail
mkdir -p gpurun_out/r06j
timeout -k 10 500 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "se_gate or benchmarked or detail" > gpurun_out/r06j/tests.log 2>&1 || { tail -30 gpurun_out/r06j/tests.log; exit 1; }
tail -3 gpurun_out/r06j/tests.log
for i in 1 2; do
  timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-inference --detail gpurun_out/r06j/new$i.json > gpurun_out/r06j/new$i.line 2> gpurun_out/r06j/new$i.err || exit 1
  EBFI_DEV=1 EBFI_NO_SEGATE_SHUFFLE=1 timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-inference --detail gpurun_out/r06j/old$i.json > gpurun_out/r06j/old$i.line 2> gpurun_out/r06j/old$i.err || exit 1
done
python - <<'PY'
import json
for t in ("new1","old1","new2","old2"):
    d=json.loads(open("gpurun_out/r06j/%s.line"%t).read().strip().splitlines()[-1])
    print(t, d["ms_per_step"], d["value"])
PY
