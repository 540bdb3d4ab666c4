#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r06o
timeout -k 10 400 python tools/convshapes.py > gpurun_out/r06o/convshapes.txt 2> gpurun_out/r06o/convshapes.err || { tail -20 gpurun_out/r06o/convshapes.err; exit 1; }
head -90 gpurun_out/r06o/convshapes.txt
