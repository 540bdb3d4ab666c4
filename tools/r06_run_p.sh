#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r06p
EBFI_DEV=1 EBFI_WGRAD_TR=0 timeout -k 10 400 python tools/convshapes.py > gpurun_out/r06p/convshapes_notr.txt 2> gpurun_out/r06p/err.txt || { tail -20 gpurun_out/r06p/err.txt; exit 1; }
grep "backward_weight" gpurun_out/r06p/convshapes_notr.txt | head -60
head -1 gpurun_out/r06p/convshapes_notr.txt
