#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r06m
timeout -k 10 400 python tools/gluestack.py > gpurun_out/r06m/gluestack.txt 2> gpurun_out/r06m/gluestack.err || { tail -20 gpurun_out/r06m/gluestack.err; exit 1; }
head -70 gpurun_out/r06m/gluestack.txt
