#!/bin/bash
# round 6, third GPU call: MFMA x FP16_OVFL NaN probe; deep vs shallow producer pipeline of conv_fwd_bf16x3_ws (kbench, same box,
# interleaved); conv parity tests on the new kernel; std probe of the benchmarked-step test; training curves with the yardstick arm
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r06c; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
echo "[1] mfma nan probe"; timeout -k 5 60 tools/bin/mfma_nan_probe > $OUT/mfma_nan_probe.log 2>&1; echo "rc=$?"; cat $OUT/mfma_nan_probe.log | cut -c1-200
echo "[2] kbench deep vs shallow"
for a in "64 64" "64 128" "128 64" "128 1600"; do
  for k in kbench kbench_shallow kbench kbench_shallow; do echo "== $k $a"; timeout -k 5 120 tools/bin/$k fwd $a 128 128 8 2>&1 | grep -E "persistent|one tile|plain, quad|max rel"; done
done > $OUT/kbench_deep_vs_shallow.log 2>&1; echo "rc=$?"; cat $OUT/kbench_deep_vs_shallow.log | grep -E "==|persistent|forward vs" 
echo "[3] conv / c16 / fac / model tests"; timeout -k 10 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_c16.py tests/test_gpu_fac.py -m gpu -x -q > $OUT/tests_conv.log 2>&1; echo "rc=$?"; tail -4 $OUT/tests_conv.log
echo "[4] std probe"; timeout -k 10 600 python tools/r06_std_probe.py > $OUT/std_probe.log 2>&1; echo "rc=$?"; grep -v "^/opt\|Warning" $OUT/std_probe.log | tail -12
echo "[5] training curves"; timeout -k 10 900 python tools/traincurves.py --steps 300 --batch 8 --size 256 --tasks copy --out $OUT/train_curves.json > $OUT/traincurves.log 2>&1; echo "rc=$?"; grep "^\[copy/[a-z0-9]*\]" $OUT/traincurves.log
timeout -k 10 300 python tools/traincurves.py --small --steps 80 --batch 2 --size 64 --lr 1e-3 --tasks copy --out $OUT/train_curves_small.json > $OUT/traincurves_small.log 2>&1; echo "rc=$?"; grep "^\[copy/[a-z0-9]*\]" $OUT/traincurves_small.log
echo "[6] bench"; timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs --no-ops --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"; cut -c1-400 $OUT/bench.json
