#!/bin/bash
# round 6, final verification: the full GPU suite, smoke, and the driver's bench form
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r06h; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
echo "[1] full GPU suite"; timeout -k 10 1800 python -m pytest tests -m gpu -x -q > $OUT/tests_all.log 2>&1; echo "rc=$?"; tail -4 $OUT/tests_all.log | cut -c1-300
echo "[2] smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "[3] driver form"; timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err; echo "rc=$?"; wc -c $OUT/bench_driver_form.json; python3 -c "
import json; d=json.loads(open('$OUT/bench_driver_form.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['inference_frames_per_s'])"
