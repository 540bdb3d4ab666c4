#!/bin/bash
# round 6, sixth GPU call: FAC backward with 8 pixels per thread (fp16 planes) vs the 4-pixel kernel; dcn_colnorm2; tests
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r06f; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
echo "[1] tests fac / dcn / model step"; timeout -k 10 900 python -m pytest tests/test_gpu_fac.py tests/test_gpu_dcn.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "rc=$?"; tail -3 $OUT/tests.log | cut -c1-300
echo "[2] facbench: 8 px (default) / 4 px"
for r in 1 2; do
timeout -k 10 120 python tools/facbench.py 2>&1 | grep "bwd in-kernel" | sed 's/^/x8  /'
EBFI_DEV=1 EBFI_FAC_BWD_X4=1 timeout -k 10 120 python tools/facbench.py 2>&1 | grep "bwd in-kernel" | sed 's/^/x4  /'
done | tee $OUT/facbench_x8_vs_x4.txt
echo "[3] opbench dcn"; timeout -k 10 200 python tools/opbench.py --ops dcn --iters 40 2>/dev/null | tail -3 | cut -c1-600
echo "[4] step tests"; timeout -k 10 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "benchmarked or reproducible or training_forward" > $OUT/tests_step.log 2>&1; echo "rc=$?"; tail -3 $OUT/tests_step.log | cut -c1-300
echo "[5] bench A/B same box: 4 px vs 8 px"
for r in 1 2; do
  EBFI_DEV=1 EBFI_FAC_BWD_X4=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-ops --no-inference --detail $OUT/ab_x4_$r.json 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('x4 round $r: %.3f ms/step' % d['ms_per_step'])"
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-ops --no-inference --detail $OUT/ab_x8_$r.json 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('x8 round $r: %.3f ms/step' % d['ms_per_step'])"
done | tee $OUT/ab_fac_bwd.txt
