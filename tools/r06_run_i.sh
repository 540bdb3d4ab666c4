#!/bin/bash
# round 6: timestamps grouped per pass in ClipInterpolator -- full GPU suite at that state, inference A/B (one timestamp per pass vs auto)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r06i; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
echo "[1] full GPU suite"; timeout -k 10 1800 python -m pytest tests -m gpu -x -q > $OUT/tests_all.log 2>&1; echo "rc=$?"; tail -4 $OUT/tests_all.log | cut -c1-300
echo "[2] inference A/B, same box: --group 1 vs default"
for r in 1 2; do
 for g in "--group 1" ""; do
  for cfg in "--batch 4 --height 256 --width 256 --precision fp32" "--batch 4 --height 256 --width 256" "--batch 1 --height 128 --width 128" "--batch 8 --height 720 --width 1280 --num_ts 8"; do
    echo -n "[$g] $cfg : "; timeout -k 10 300 python ebfi-be_amd/infer_ours.py --rand-init $cfg $g 2>/dev/null | tail -1 | cut -c1-110
  done
 done
done | tee $OUT/infer_group_ab.txt
