#!/bin/bash
# round 6, batch k: which PyTorch-native kernels are left in the replayed step, by launch size
set -o pipefail
OUT=gpurun_out/r06k
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o g -- python3 tools/graphprof.py --steps 12 > $OUT/graphprof.log 2>&1 || { tail -5 $OUT/graphprof.log; exit 1; }
T=$(find $OUT/trace -name '*kernel_trace.csv' | sort | tail -1)
N=$(python3 - $T <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_flat" in r["Kernel_Name"]]
print(idx[-1] - idx[-2])
PY
)
echo "trace $T per-step launches: $N"
tail -3 $OUT/graphprof.log
python3 tools/gluetrace.py $T ${N:-563} > $OUT/glue.txt && cat $OUT/glue.txt
rm -rf $OUT/trace
