#!/bin/bash
# round 6, fourth GPU call: FP16_OVFL fix + fp16 fused inference kernel: targeted tests, full suite, training curves (fixed yardstick), bench
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r06d; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
echo "[1] targeted tests"; timeout -k 10 900 python -m pytest tests/test_gpu_c16.py tests/test_gpu_fac.py tests/test_gpu_training_curve.py tests/test_infer_cli.py -m gpu -x -q > $OUT/tests_new.log 2>&1; echo "rc=$?"; tail -12 $OUT/tests_new.log | cut -c1-300
echo "[2] bench"; timeout -k 10 600 python bench.py --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"; cut -c1-300 $OUT/bench.json; grep "inference config" $OUT/bench.err
echo "[2b] bench inference A/B: split-precision fused kernel"; EBFI_DEV=1 EBFI_NO_FAC_F16=1 timeout -k 10 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --no-ops --detail $OUT/bench_nof16_detail.json > $OUT/bench_nof16.json 2> $OUT/bench_nof16.err; echo "rc=$?"; grep "inference config" $OUT/bench_nof16.err
echo "[3] full GPU suite"; timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $OUT/tests_all.log 2>&1; echo "rc=$?"; tail -6 $OUT/tests_all.log | cut -c1-300
echo "[4] training curves"; timeout -k 10 900 python tools/traincurves.py --steps 300 --batch 8 --size 256 --tasks copy --out $OUT/train_curves.json > $OUT/traincurves.log 2>&1; echo "rc=$?"; grep "^\[copy/[a-z0-9]*\]" $OUT/traincurves.log
