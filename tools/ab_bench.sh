#!/bin/bash
# Same-box A/B of bench.py configurations (development): each line of output = one configuration's headline numbers.
# usage (GPU box): tools/ab_bench.sh "NAME=ENV1=v ENV2=v" ...   e.g.  tools/ab_bench.sh "prev=EBFI_LIB_PATH=$PWD/ebfi-be_amd/lib/libebfi_hip_prev.so EBFI_NO_BANK=1" "new="
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/ab
for round in 1 2; do
for spec in "$@"; do
  name="${spec%%=*}"; envs="${spec#*=}"
  env EBFI_DEV=1 $envs timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs --no-ops --no-inference --detail gpurun_out/ab/$name.$round.detail.json > gpurun_out/ab/$name.$round.json 2> gpurun_out/ab/$name.$round.err || { echo "$name failed"; tail -3 gpurun_out/ab/$name.$round.err; continue; }
  python3 - "$name" "$round" <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab/%s.%s.detail.json"%(sys.argv[1],sys.argv[2])))
k=d["kernels"]
top=sorted(k.items(), key=lambda kv:-kv[1]["total_ms"])[:4]
print("%-12s round %s: %7.3f ms/step  %s" % (sys.argv[1], sys.argv[2], d["ms_per_step"], "  ".join("%s %.1fx%.4f"%(n,v["launches_per_step"],v["avg_ms"]) for n,v in top)), flush=True)
PY
done; done
