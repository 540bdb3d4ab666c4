#!/bin/bash
# round 6, batch q: thin weight gradients on the matrix cores (conv2d_shift.inc.hpp) -- tests, then same-box A/B of the step
set -o pipefail
mkdir -p gpurun_out/r06q
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "thin_layer" > gpurun_out/r06q/tests.log 2>&1 || { tail -40 gpurun_out/r06q/tests.log; exit 1; }
tail -3 gpurun_out/r06q/tests.log
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r06q/tests2.log 2>&1 || { tail -40 gpurun_out/r06q/tests2.log; exit 1; }
tail -3 gpurun_out/r06q/tests2.log
for i in 1 2; do
  timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-inference --detail gpurun_out/r06q/new$i.json > gpurun_out/r06q/new$i.line 2> gpurun_out/r06q/new$i.err || exit 1
  EBFI_DEV=1 EBFI_NO_SHIFT_WGRAD=1 timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-inference --detail gpurun_out/r06q/old$i.json > gpurun_out/r06q/old$i.line 2> gpurun_out/r06q/old$i.err || exit 1
done
python - <<'PY'
import json
for t in ("new1","old1","new2","old2"):
    d=json.loads(open("gpurun_out/r06q/%s.line"%t).read().strip().splitlines()[-1])
    print(t, d["ms_per_step"], d["value"])
a=json.load(open("gpurun_out/r06q/new1.json"))["kernels"]; b=json.load(open("gpurun_out/r06q/old1.json"))["kernels"]
for k in sorted(set(a)|set(b)):
    x=a.get(k,{}).get("total_ms",0)/30; y=b.get(k,{}).get("total_ms",0)/30
    if abs(x-y)>0.01: print("%-36s new %.4f old %.4f d %+.4f"%(k,x,y,x-y))
print({r:(v["launches"]//30, round(v["total_ms"]/30,4)) for r,v in a.get("conv_wgrad_shift",{}).get("roles",{}).items()})
PY
