#!/bin/bash
# Runs on the GPU box (gpurun): the measurements that profiles/<round>/ is built from.  Output: gpurun_out/final/.
# One GPU step after the other; a failing step stops the script (no retries).
set -eo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/${1:-final}
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
echo "[1] smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
echo "[2] bench (default, with cpu baseline)"; timeout -k 10 600 python bench.py --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err; wc -c $OUT/bench.json; cut -c1-160 $OUT/bench.json
echo "[3] rocprofv3 kernel stats of the bench command"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --no-inference --no-ops --detail $OUT/bench_prof_detail.json > $OUT/bench_prof.json 2> $OUT/bench_prof.err
find $OUT/prof -name "*kernel_trace*" -delete; find $OUT/prof -name "*.csv" | head
echo "[3b] timeline of graph-replayed steps"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o g -- python3 tools/graphprof.py --steps 12 > $OUT/graphprof.log 2>&1
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1); python3 tools/trace_summary.py $f auto > $OUT/graph_replay_timeline.txt; rm -rf $OUT/trace; head -5 $OUT/graph_replay_timeline.txt
echo "[4] PMC passes (eager launches), FETCH_SIZE then WRITE_SIZE"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --no-inference --no-ops --no-graph --detail /tmp/pmc_detail.json > /dev/null 2> $OUT/pmc_$C.err
done
python3 tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/pmc_traffic_summary.json; cat $OUT/pmc_traffic_summary.json | head -30
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
echo "[5] op microbenchmarks"
timeout -k 10 300 python tools/opbench.py --ops fac,dcn,conv --x3 --iters 20 > $OUT/opbench_x3.jsonl 2> $OUT/opbench_x3.err
echo "[5b] kernel harness (in-kernel stamps of the forward kernel, weight gradient)"
( for a in "64 64" "128 64" "64 128" "128 1600"; do timeout -k 5 60 tools/bin/kbench fwd $a 128 128 8; timeout -k 5 60 tools/bin/kbench wgrad $a 128 128 8; done ) > $OUT/kbench.log 2>&1 || true
timeout -k 10 300 python tools/opbench.py --ops conv --iters 20 > $OUT/opbench_fp32.jsonl 2> $OUT/opbench_fp32.err
timeout -k 10 300 python tools/opbench.py --ops fac,dcn --iters 40 > $OUT/opbench_dcn_fac.jsonl 2> $OUT/opbench_dcn_fac.err; tail -1 $OUT/opbench_dcn_fac.jsonl | cut -c1-220
echo "[5a] rocprofv3 kernel stats of the op block (the north_star DCNv2+FAC figure needs a rocprof line behind the hipEvent pairs)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ops -o ops -- python3 tools/opbench.py --ops fac,dcn --iters 40 > $OUT/opbench_dcn_fac_prof.jsonl 2> $OUT/opbench_dcn_fac_prof.err
find $OUT/prof_ops -name "*kernel_trace*" -delete; find $OUT/prof_ops -name "*kernel_stats.csv" | head -2
echo "[5c] backward kernels per shape, split-precision vs fp16"
timeout -k 10 300 python tools/f16bench.py > $OUT/f16bench.log 2>&1 || true; tail -3 $OUT/f16bench.log
echo "[6] other BASELINE configs (re-randomised weights: the x0.1 initialisation gives a constant output)"
timeout -k 10 300 python ebfi-be_amd/infer_ours.py --rand-init --batch 1 --height 128 --width 128 > $OUT/config1.log 2>&1; tail -1 $OUT/config1.log
timeout -k 10 300 python ebfi-be_amd/infer_ours.py --rand-init --batch 4 --height 256 --width 256 --precision fp32 > $OUT/config2_fp32.log 2>&1; tail -1 $OUT/config2_fp32.log
timeout -k 10 300 python ebfi-be_amd/infer_ours.py --rand-init --batch 4 --height 256 --width 256 > $OUT/config2_x3.log 2>&1; tail -1 $OUT/config2_x3.log
timeout -k 10 400 python ebfi-be_amd/infer_ours.py --rand-init --batch 8 --height 720 --width 1280 --num_ts 4 > $OUT/config5.log 2>&1; tail -1 $OUT/config5.log
EBFI_DEV=1 EBFI_NO_FAC_FUSION=1 timeout -k 10 400 python ebfi-be_amd/infer_ours.py --rand-init --batch 8 --height 720 --width 1280 --num_ts 4 > $OUT/config5_unfused.log 2>&1; tail -1 $OUT/config5_unfused.log
# (round 6: the fused kernel runs on fp16 operands by default; the split-precision form of it for comparison)
EBFI_DEV=1 EBFI_NO_FAC_F16=1 timeout -k 10 400 python ebfi-be_amd/infer_ours.py --rand-init --batch 8 --height 720 --width 1280 --num_ts 4 > $OUT/config5_fused_x3.log 2>&1; tail -1 $OUT/config5_fused_x3.log
EBFI_DEV=1 EBFI_NO_FAC_F16=1 timeout -k 10 300 python ebfi-be_amd/infer_ours.py --rand-init --batch 4 --height 256 --width 256 > $OUT/config2_fused_x3.log 2>&1; tail -1 $OUT/config2_fused_x3.log
echo "[6b] rocprofv3 kernel stats of the inference configs 2 and 5 (eager launches: a replayed graph is traced the same way)"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c2 -o c2 -- python3 ebfi-be_amd/infer_ours.py --rand-init --batch 4 --height 256 --width 256 --num_ts 8 > $OUT/config2_prof.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5 -o c5 -- python3 ebfi-be_amd/infer_ours.py --rand-init --batch 8 --height 720 --width 1280 --num_ts 4 > $OUT/config5_prof.log 2>&1
find $OUT/prof_c2 $OUT/prof_c5 -name "*kernel_trace*" -delete; find $OUT/prof_c2 $OUT/prof_c5 -name "*kernel_stats.csv" | head
echo "[6c] train_ours.py steady-state rate (device-side synthetic batches, hipGraph replay)"
( cd ebfi-be_amd && timeout -k 10 300 python train_ours.py --graph --iterations 120 ) > $OUT/train_ours_1gpu.log 2>&1; tail -2 $OUT/train_ours_1gpu.log
echo "[7] two-rank rehearsal of the bench (both ranks on this GPU, gloo)"
EBFI_BENCH_REHEARSAL=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --detail $OUT/bench_2rank_rehearsal_detail.json > $OUT/bench_2rank_rehearsal.out 2> $OUT/bench_2rank_rehearsal.err; grep '^{' $OUT/bench_2rank_rehearsal.out > $OUT/bench_2rank_rehearsal.json; cut -c1-160 $OUT/bench_2rank_rehearsal.json   # (gloo prints a connection banner on stdout)
echo "[8] fp16 operand storage: readers / writers of the images in isolation; the detail branch by kernel"
timeout -k 10 200 python tools/c16bench.py > $OUT/c16bench.log 2>&1 || true; tail -4 $OUT/c16bench.log
timeout -k 10 200 python tools/detailprof.py > $OUT/detailprof.log 2>&1 || true; head -3 $OUT/detailprof.log
timeout -k 10 200 python tools/detail_layers.py > $OUT/detail_layers.txt 2> $OUT/detail_layers.err || true
du -sh $OUT
