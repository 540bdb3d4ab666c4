#!/bin/bash
# round 6, seventh GPU call: FAC backward with the scales out of the inner loop; full GPU suite at the candidate final state
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r06g; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
echo "[1] facbench"; for r in 1 2; do
timeout -k 10 120 python tools/facbench.py 2>&1 | grep "bwd in-kernel" | sed 's/^/x8 scales out  /'
EBFI_DEV=1 EBFI_FAC_BWD_X4=1 timeout -k 10 120 python tools/facbench.py 2>&1 | grep "bwd in-kernel" | sed 's/^/x4             /'
done | tee $OUT/facbench_x8s_vs_x4.txt
echo "[2] full GPU suite"; timeout -k 10 1800 python -m pytest tests -m gpu -x -q > $OUT/tests_all.log 2>&1; echo "rc=$?"; tail -6 $OUT/tests_all.log | cut -c1-300
echo "[3] smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
