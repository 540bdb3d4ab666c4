#!/bin/bash
# round 6, first GPU call: the new entry-point / bench-line tests, the non-finite-input probe, one default bench run
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r06a; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
echo "[1] nanprobe"; timeout -k 10 200 python tools/nanprobe.py > $OUT/nanprobe.log 2>&1; echo "rc=$?"; tail -5 $OUT/nanprobe.log
echo "[2] tests"; timeout -k 10 900 python -m pytest tests/test_infer_cli.py tests/test_clipdata.py tests/test_gpu_entrypoints.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "rc=$?"; tail -5 $OUT/tests.log
echo "[3] bench"; timeout -k 10 600 python bench.py --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"; wc -c $OUT/bench.json; cat $OUT/bench.json
