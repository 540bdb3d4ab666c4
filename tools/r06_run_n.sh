#!/bin/bash
# round 6, batch n: the whole GPU suite + smoke at the current tree
set -o pipefail
mkdir -p gpurun_out/r06n
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r06n/tests.log 2>&1 || { tail -40 gpurun_out/r06n/tests.log; exit 1; }
tail -3 gpurun_out/r06n/tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06n/smoke.log 2>&1 || { tail -20 gpurun_out/r06n/smoke.log; exit 1; }
tail -2 gpurun_out/r06n/smoke.log
