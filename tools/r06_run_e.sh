#!/bin/bash
# round 6, fifth GPU call: same-box A/B of the training step, round-5 final tree (_prev/, a git worktree of 67f55a7) against the
# current tree, alternating; fused fp16 kernel with image input: tests + inference legs
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/r06e; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
echo "[1] fac tests"; timeout -k 10 600 python -m pytest tests/test_gpu_fac.py tests/test_infer_cli.py -m gpu -x -q > $OUT/tests_fac.log 2>&1; echo "rc=$?"; tail -3 $OUT/tests_fac.log | cut -c1-300
echo "[2] same-box A/B: r05 final vs now (ms per step, 20 steps each, alternating)"
for round in 1 2 3; do
  ( cd _prev && timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-ops --no-inference 2> /dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('r05  round $round: %.3f ms/step' % d['ms_per_step'])" )
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-ops --no-inference --detail $OUT/ab_now_$round.json 2> /dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('now  round $round: %.3f ms/step' % d['ms_per_step'])"
done 2>&1 | tee $OUT/ab_r05_vs_now.txt
echo "[3] inference legs"; timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --no-ops --detail $OUT/bench_inf_detail.json > $OUT/bench_inf.json 2> $OUT/bench_inf.err; echo "rc=$?"; grep "inference config" $OUT/bench_inf.err
EBFI_DEV=1 EBFI_NO_FAC_IMG=1 timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --no-ops --detail $OUT/bench_inf_noimg_detail.json > $OUT/bench_inf_noimg.json 2> $OUT/bench_inf_noimg.err; echo "rc=$?"; grep "inference config" $OUT/bench_inf_noimg.err
