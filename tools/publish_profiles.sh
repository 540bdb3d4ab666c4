#!/bin/bash
# Copies the summaries of a tools/collect_profiles.sh run (gpurun_out/<run>/) into profiles/<round>/ under the *_final names.
# usage: tools/publish_profiles.sh <run dir under gpurun_out> <round dir under profiles>
set -euo pipefail
cd "$(dirname "$0")/.."
S=gpurun_out/$1; D=profiles/$2
mkdir -p "$D"
cp $S/bench.json $D/bench_final.json
cp $S/bench_prof.json $D/bench_final_profiled_run.json
cp $S/prof/bench_kernel_stats.csv $D/bench_final_kernel_stats.csv
cp $S/graph_replay_timeline.txt $D/graph_replay_timeline_final.txt
cp $S/pmc_traffic_summary.json $D/bench_final_pmc_traffic_summary.json
cp $S/kbench.log $D/kbench_final.log
cp $S/opbench_x3.jsonl $D/opbench_final_bf16x3.jsonl
cp $S/opbench_fp32.jsonl $D/opbench_final_fp32.jsonl
cp $S/opbench_dcn_fac.jsonl $D/opbench_dcn_fac_final.jsonl
cat $S/config1.log $S/config2_fp32.log $S/config2_x3.log $S/config5.log | grep -v amdgpu.ids > $D/configs_final.log
cp $S/bench_2rank_rehearsal.json $D/bench_final_2rank_rehearsal.json
grep -v amdgpu.ids $S/smoke.log > $D/smoke_final.log
echo "published $S -> $D"
