#!/bin/bash
# round 6, second GPU call: non-finite probe on the wave-specialised kernel, scalar-conv bank, training curves, full GPU suite
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r06b; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
echo "[1] nanprobe"; timeout -k 10 300 python tools/nanprobe.py > $OUT/nanprobe.log 2>&1; echo "rc=$?"; grep -c DIFFER $OUT/nanprobe.log; tail -2 $OUT/nanprobe.log
echo "[2] new tests"; timeout -k 10 900 python -m pytest tests/test_gpu_scalar_conv.py tests/test_gpu_training_curve.py -m gpu -x -q > $OUT/tests_new.log 2>&1; echo "rc=$?"; tail -15 $OUT/tests_new.log
echo "[3] training curves, full size"; timeout -k 10 600 python tools/traincurves.py --steps 300 --batch 8 --size 256 --out $OUT/train_curves.json > $OUT/traincurves.log 2>&1; echo "rc=$?"; tail -8 $OUT/traincurves.log
echo "[4] full GPU suite"; timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $OUT/tests_all.log 2>&1; echo "rc=$?"; tail -8 $OUT/tests_all.log
