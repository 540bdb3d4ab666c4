#!/bin/bash
# Copies the summaries of a tools/collect_profiles.sh run (gpurun_out/<run>/) into profiles/<round>/ under the *_final names.
# usage: tools/publish_profiles.sh <run dir under gpurun_out> <round dir under profiles>
set -euo pipefail
cd "$(dirname "$0")/.."
S=gpurun_out/$1; D=profiles/$2
mkdir -p "$D"
cp $S/bench.json $D/bench_final.json
[ -s $S/bench_detail.json ] && cp $S/bench_detail.json $D/bench_final_detail.json || true      # (round 6: the line is compact, the tables are here)
[ -s $S/bench_prof_detail.json ] && cp $S/bench_prof_detail.json $D/bench_final_profiled_run_detail.json || true
[ -s $S/bench_2rank_rehearsal_detail.json ] && cp $S/bench_2rank_rehearsal_detail.json $D/bench_final_2rank_rehearsal_detail.json || true
f=$(find $S/prof_ops -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp $f $D/opbench_dcn_fac_kernel_stats.csv || true
cp $S/bench_prof.json $D/bench_final_profiled_run.json
cp $S/prof/bench_kernel_stats.csv $D/bench_final_kernel_stats.csv
cp $S/graph_replay_timeline.txt $D/graph_replay_timeline_final.txt
cp $S/pmc_traffic_summary.json $D/bench_final_pmc_traffic_summary.json
# the figures bench.py quotes as `traffic` / `traffic_ratio`: the summary of the latest published collection
python3 - "$S/pmc_traffic_summary.json" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
d = {"_comment": "HBM-side bytes per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over `bench.py --no-graph`, "
                 "mean over all launches of a kernel; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes -- calibrated here for "
                 "coalesced 4-, 8- and 16-byte-per-lane reads alike (profiles/r05/fetch_size_calibration.txt); the DCNv2 gather kernels "
                 "are reported as counted: fetch_correction); source: profiles/%s/bench_final_pmc_traffic_summary.json" % sys.argv[2], **d}
json.dump(d, open("profiles/pmc_traffic.json", "w"), indent=1)
PY
cp $S/kbench.log $D/kbench_final.log
cp $S/opbench_x3.jsonl $D/opbench_final_bf16x3.jsonl
cp $S/opbench_fp32.jsonl $D/opbench_final_fp32.jsonl
cp $S/opbench_dcn_fac.jsonl $D/opbench_dcn_fac_final.jsonl
( for f in config1 config2_fp32 config2_x3 config2_fused_x3 config5 config5_fused_x3 config5_unfused; do [ -s $S/$f.log ] && { echo "== $f"; grep -v amdgpu.ids $S/$f.log; }; done ) > $D/configs_final.log
cp $(find $S/prof_c2 -name "*kernel_stats.csv" | head -1) $D/infer_config2_kernel_stats.csv
cp $(find $S/prof_c5 -name "*kernel_stats.csv" | head -1) $D/infer_config5_kernel_stats.csv
grep -v amdgpu.ids $S/train_ours_1gpu.log > $D/train_ours_1gpu.log
grep -v amdgpu.ids $S/f16bench.log > $D/f16bench_final.log || true
cp $S/bench_2rank_rehearsal.json $D/bench_final_2rank_rehearsal.json
grep -v amdgpu.ids $S/smoke.log > $D/smoke_final.log
grep -v amdgpu.ids $S/c16bench.log > $D/c16bench_final.log || true
grep -v "amdgpu.ids\|Warning\|_warn_once" $S/detailprof.log > $D/detail_branch_by_kernel.txt || true
[ -s $S/detail_layers.txt ] && cp $S/detail_layers.txt $D/detail_branch_by_stage_latest.txt || true
echo "published $S -> $D"
